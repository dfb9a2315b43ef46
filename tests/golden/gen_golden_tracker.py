#!/usr/bin/env python3
"""Golden vectors of the reference regression-tracking receiver (LEGACY/gr-ofdm-rx/python/SynchronizeAndEstimate.py), build
container only:

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_tracker.py   -> tests/golden/ref_tracker.npz

The file is Python-3 code and is executed unmodified in-process with the usual shims (stub gnuradio.gr.sync_block,
np.product alias).  Nothing is written to /root/reference; only arrays are saved.  Inputs: oracle TX (N=64, cp=16, 62 sync /
60 data bins, [1,3] pattern, ZC root 23 = what the block's case 0 expects), clean / reference 5-tap channel / AWGN.
"""
import contextlib
import io
import os
import warnings

import numpy as np

import gen_golden_fo as G            # installs the gnuradio stub
from oracle import ofdm_oracle as orc

np.product = np.prod
PATH = "/root/reference/GNU-Radio-Repositories/LEGACY/gr-ofdm-rx/python/SynchronizeAndEstimate.py"

CASES = [
    # tag, block case, symbols sent, lead samples, fading, noise sigma
    ("clean", 0, 48, 7, False, 0.0),
    ("fade", 0, 48, 21, True, 0.0),
    ("noisy", 0, 44, 2, True, 0.02),
    ("mimo_cfg", 1, 12, 5, True, 0.0),
    # the reference's own TX fixture (LEGACY/gr-ofdm-tx/python/tx_data_0.pckl: 48 symbols x 80 samples, [1,3] pattern, ZC root 23),
    # read with the non-executing ndarray-pickle parser; no bit fixture exists for it
    ("txdata0", 0, None, 0, False, 0.0),
]
FIXTURE = "/root/reference/GNU-Radio-Repositories/LEGACY/gr-ofdm-tx/python/tx_data_0.pckl"


def make_input(n_sym, lead, fading, sigma, seed):
    rng = np.random.default_rng(seed)
    n_data = sum(1 for s in range(n_sym) if s % 4 >= 1)
    bits = rng.integers(0, 2, n_data * 60 * 2)
    tx = orc.tx_modulate(bits, 64, 16, 62, 60, n_sym, synch_dat=(1, 3), zc_root=23)
    if fading:
        tx = orc.channel_apply(tx, orc.REF_TAPS, 64)[:len(tx) + 8]
    tx = tx + sigma * (rng.standard_normal(len(tx)) + 1j * rng.standard_normal(len(tx)))
    return np.concatenate([np.zeros(lead), tx, np.zeros(40)]).astype(np.complex64), bits


def main():
    ns = {"__name__": "ref_synchronize_and_estimate"}
    exec(compile(open(PATH).read(), PATH, "exec"), ns)
    cls = ns["SynchronizeAndEstimate"]
    out = {}
    warnings.simplefilter("ignore")          # FutureWarning of np.linalg.lstsq's default rcond
    for i, (tag, case, n_sym, lead, fading, sigma) in enumerate(CASES):
        if n_sym is None:
            import sys
            sys.path.insert(0, os.path.join(G.ROOT, "lte-gnu-radio-code_amd"))
            from ofdm_mi355x.safe_pickle import load_ndarray
            iq = np.concatenate([load_ndarray(FIXTURE)[0], np.zeros(40)]).astype(np.complex64)
            bits = np.zeros(0, np.uint8)
        else:
            iq, bits = make_input(n_sym, lead, fading, sigma, 700 + i)
        blk = cls(case)
        out[tag + "_iq"] = iq
        out[tag + "_bits"] = bits.astype(np.uint8)
        out[tag + "_case"] = np.array([case])
        for call in (1, 2):
            o = np.zeros(len(iq), np.complex64)
            with contextlib.redirect_stdout(io.StringIO()):          # SE:439 prints a shape per data symbol
                blk.work([iq], [o])
            print(tag, "call", call, "corr_obs", blk.corr_obs, blk.time_synch_ref[0, :8, 0].tolist(), blk.time_synch_ref[0, :8, 1].tolist())
            k = "%s_call%d_" % (tag, call)
            out[k + "corr_obs"] = np.array([blk.corr_obs])
            out[k + "tsr"] = blk.time_synch_ref.copy()
            out[k + "Hp"] = blk.est_chan_freq_p.copy()
            out[k + "Hn"] = blk.est_chan_freq_n.copy()
            out[k + "imp"] = blk.est_chan_impulse.copy()
            out[k + "esf"] = blk.est_synch_freq.copy()
            if case == 0:
                out[k + "edf"] = blk.est_data_freq.copy()
            out[k + "out"] = o
    np.savez_compressed(os.path.join(G.HERE, "ref_tracker.npz"), **out)


if __name__ == "__main__":
    main()
