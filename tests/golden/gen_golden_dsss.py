#!/usr/bin/env python3
"""Golden vectors of the reference DSSS receiver (LEGACY/gr-ofdm-rx/python/SynchEstFOAndDSSS.py), build container only:

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_dsss.py      -> tests/golden/ref_dsss.npz

Same method as gen_golden_fo.py (the file is Python-2 code and is run in-process through the Python-2 division transform);
only arrays are saved.  Inputs are synthetic: QPSK symbols spread over DSSS consecutive bins with the block's own code
(oracle.dsss_spread; the reference has no transmitter for this mode), ZC root 37 sync symbols, small carrier offsets.
"""
import os

import numpy as np

import gen_golden_fo as G
from oracle import ofdm_oracle as orc

PATH = "/root/reference/GNU-Radio-Repositories/LEGACY/gr-ofdm-rx/python/SynchEstFOAndDSSS.py"

CASES = [
    # tag, case, fo_range (Hz), true carrier offset (Hz), lead samples, fading
    ("d1", 1, [-1500, 0, 1500], 200.0, 2, False),
    ("d3", 3, [0, 1000], -300.0, 0, True),
    ("d4", 4, [-2000, 0, 2000], 400.0, 5, True),
    ("d8", 8, [0], 0.0, 9, False),
    ("d10", 10, [-3000, 3000], 250.0, 1, True),
]


def make_input(case, cfo_hz, lead, fading, seed):
    n_symb, fs, N, sd, Kd, dsss = orc.DSSS_CASES[case]
    S, D = sd
    cp = N // 4
    rng = np.random.default_rng(seed)
    n_data = sum(1 for s in range(n_symb) if s % (S + D) >= S)
    n_spread = Kd // dsss
    bits = rng.integers(0, 2, n_data * n_spread * 2)
    sym = orc.map_bits(bits, "QPSK").reshape(n_data, n_spread)
    tx = orc.tx_modulate(None, N, cp, N - 2, Kd, n_symb, synch_dat=(S, D), zc_root=37, zc_segments=True,
                         zc_parity_of_bins=True, data_symbols=orc.dsss_spread(sym, dsss, Kd))
    if fading:
        tx = orc.channel_apply(tx, orc.REF_TAPS, N)[:len(tx) + 8]
    rx = tx * np.exp(1j * 2 * np.pi * cfo_hz / fs * np.arange(len(tx)))
    return np.concatenate([np.zeros(lead), rx, np.zeros(2 * cp)]).astype(np.complex64), bits


def main():
    cls = G.load_reference_class(PATH, "SynchEstFOAndDSSS")
    out = {}
    for i, (tag, case, fo_range, cfo_hz, lead, fading) in enumerate(CASES):
        iq, bits = make_input(case, cfo_hz, lead, fading, 300 + i)
        blk = cls(case, fo_range, "/tmp/", "x", 0)
        out[tag + "_iq"] = iq
        out[tag + "_bits"] = bits.astype(np.uint8)
        out[tag + "_fo_range"] = np.array(fo_range, dtype=np.float64)
        out[tag + "_meta"] = np.array([case, cfo_hz, lead], dtype=np.float64)
        for call in (1, 2):
            o = np.zeros(len(iq), np.complex64)
            blk.work([iq], [o])
            n_found = int(np.count_nonzero(blk.time_synch_ref[:, 2]))
            print(tag, "call", call, "syncs", n_found, "fo idx", blk.dmax_tmp_ind, "first", blk.time_synch_ref[:3, :2].tolist())
            k = "%s_call%d_" % (tag, call)
            out[k + "tsr"] = blk.time_synch_ref.copy()
            out[k + "fo_idx"] = np.array([blk.dmax_tmp_ind])
            out[k + "H"] = blk.est_chan_freq_P.copy()
            out[k + "edf"] = blk.est_data_freq.copy()
            out[k + "edfd"] = blk.est_data_freq_d.copy()
            out[k + "out"] = o
    np.savez_compressed(os.path.join(G.HERE, "ref_dsss.npz"), **out)


if __name__ == "__main__":
    main()
