#!/usr/bin/env python3
"""Golden vectors of the reference CFO-search receiver (LEGACY/gr-ofdm-rx/python/SynchEstAndFO.py), build container only:

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_fo.py      -> tests/golden/ref_fo.npz

The file is Python-2 era: `self.cp_len = self.nfft/4` and `corr_size = num_ofdm_symb/sum(synch_dat)` rely on integer `/`.
It is executed in-process from its own source through an AST transform that gives every `/` its Python-2 meaning (floor
division when both operands are ints, true division otherwise), next to the usual shims (np.complex alias, stub
gnuradio.gr.sync_block).  Nothing is written to /root/reference and no reference source is copied: only arrays are saved.
Inputs are synthetic (oracle TX with ZC root 37 / per-segment sync symbols, a carrier offset, the reference 5-tap channel).
"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ofdm_oracle as orc  # noqa: E402

np.complex = complex
_gr = types.ModuleType("gnuradio")
_grgr = types.ModuleType("gnuradio.gr")


class _sync_block:
    def __init__(self, name=None, in_sig=None, out_sig=None):
        pass


_grgr.sync_block = _sync_block
_gr.gr = _grgr
sys.modules["gnuradio"] = _gr
sys.modules["gnuradio.gr"] = _grgr

PATH = "/root/reference/GNU-Radio-Repositories/LEGACY/gr-ofdm-rx/python/SynchEstAndFO.py"


class _Py2Div(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()), args=[node.left, node.right], keywords=[]), node)
        return node


def _py2div(a, b):
    ints = (int, np.integer)
    if isinstance(a, ints) and isinstance(b, ints):
        return a // b
    return a / b


def load_reference_class(path=PATH, name="SynchEstAndFO"):
    tree = ast.fix_missing_locations(_Py2Div().visit(ast.parse(open(path).read())))
    ns = {"_py2div": _py2div, "__name__": "ref_" + name}
    exec(compile(tree, path, "exec"), ns)
    return ns[name]


CASES = [
    # tag, case, fo_range (Hz), true carrier offset (Hz), lead samples, fading
    # (under the file's Python-2 semantics 1/fs == 0: its rotators are all ones, so small true offsets are used)
    ("c0", 0, [-3000, -1500, 0, 1500, 3000], -400.0, 5, False),
    ("c3", 3, [-2000, 0, 2000], 300.0, 3, True),
    ("c6", 6, [-4000, -2000, 0, 2000, 4000], 500.0, 0, True),
    ("c9", 9, [-6000, 0, 6000], -300.0, 7, False),
]


def make_input(case, cfo_hz, lead, fading, seed):
    n_symb, fs, N, sd, Kd = orc.FO_CASES[case]
    S, D = sd
    cp = N // 4
    rng = np.random.default_rng(seed)
    n_sym = n_symb
    n_data = sum(1 for s in range(n_sym) if s % (S + D) >= S)
    bits = rng.integers(0, 2, n_data * Kd * 2)
    tx = orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym, synch_dat=(S, D), zc_root=37, zc_segments=True, zc_parity_of_bins=True)
    if fading:
        tx = orc.channel_apply(tx, orc.REF_TAPS, N)[:len(tx) + 8]
    n = np.arange(len(tx))
    rx = tx * np.exp(1j * 2 * np.pi * cfo_hz / fs * n)
    return np.concatenate([np.zeros(lead), rx, np.zeros(2 * cp)]).astype(np.complex64), bits


def main():
    cls = load_reference_class()
    out = {}
    for i, (tag, case, fo_range, cfo_hz, lead, fading) in enumerate(CASES):
        iq, bits = make_input(case, cfo_hz, lead, fading, 100 + i)
        blk = cls(case, fo_range, "/tmp/", "x", 0)
        out[tag + "_iq"] = iq
        out[tag + "_fo_range"] = np.array(fo_range, dtype=np.float64)
        out[tag + "_meta"] = np.array([case, cfo_hz, lead], dtype=np.float64)
        for call in (1, 2):
            o = np.zeros(len(iq), np.complex64)
            blk.work([iq], [o])
            n_found = int(np.count_nonzero(blk.time_synch_ref[:, 2]))
            print(tag, "call", call, "syncs", n_found, "fo idx", blk.dmax_tmp_ind, "first", blk.time_synch_ref[:3, :2].tolist())
            out["%s_call%d_tsr" % (tag, call)] = blk.time_synch_ref.copy()
            out["%s_call%d_fo_idx" % (tag, call)] = np.array([blk.dmax_tmp_ind])
            out["%s_call%d_H" % (tag, call)] = blk.est_chan_freq_P.copy()
            out["%s_call%d_htime" % (tag, call)] = blk.est_chan_time.copy()
            out["%s_call%d_edf" % (tag, call)] = blk.est_data_freq.copy()
            out["%s_call%d_esf" % (tag, call)] = blk.est_synch_freq.copy()
            out["%s_call%d_out" % (tag, call)] = o
    np.savez_compressed(os.path.join(HERE, "ref_fo.npz"), **out)


if __name__ == "__main__":
    main()
