#!/usr/bin/env python3
"""Golden vectors of gr-RXOFDM's synch_and_chan_est.work ("table mode": up to 100 syncs per call, one data symbol per
sync), build container only:

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_rxofdm_table.py  -> tests/golden/ref_rxofdm_table.npz

As shipped the class cannot get past its first detection: synch_and_chan_est.py:194 reads `self.diagnostic`, an attribute the
constructor never sets (it sets `self.diagnostics`), and :253 relies on Python-2 integer `/`.  To record what its arithmetic
does, the instance is given `diagnostic = 0` from outside after construction and the file is run through the same
Python-2 division transform as gen_golden_fo.py.  Nothing is written to /root/reference; only arrays are saved.
"""
import contextlib
import io
import os

import numpy as np

import gen_golden_fo as G
from oracle import ofdm_oracle as orc

PATH = "/root/reference/GNU-Radio-Repositories/gr-RXOFDM/python/synch_and_chan_est.py"
# the LEGACY module's block of the same name has the identical work() without the broken genie branch and runs unmodified
PATH_LEGACY = "/root/reference/GNU-Radio-Repositories/LEGACY/gr-ofdm-rx/python/SynchAndChanEst.py"

CASES = [
    # tag, (num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr), n_sym sent, lead, fading
    ("chain", (24, 64, 16, 64, [1, 3], 60, 50), 24, 3, True),          # ofdm_chain.py:83 wiring (K = N sync bins)
    ("s2", (48, 128, 32, 126, [2, 1], 100, 1e8), 48, 0, False),
    ("n256", (12, 256, 64, 254, [1, 1], 180, 1000.0), 12, 11, True),
]


def make_input(par, n_sym, lead, fading, seed):
    _, N, cp, Ks, sd, Kd, _ = par
    S, D = sd
    rng = np.random.default_rng(seed)
    n_data = sum(1 for s in range(n_sym) if s % (S + D) >= S)
    bits = rng.integers(0, 2, n_data * Kd * 2)
    tx = orc.tx_modulate(bits, N, cp, Ks, Kd, n_sym, synch_dat=(S, D), zc_root=37, zc_segments=True, zc_parity_of_bins=True)
    if fading:
        tx = orc.channel_apply(tx, orc.REF_TAPS, N)[:len(tx) + 8]
    return np.concatenate([np.zeros(lead), tx, np.zeros(2 * cp)]).astype(np.complex64), bits


def main():
    cls = G.load_reference_class(PATH, "synch_and_chan_est")
    cls_legacy = G.load_reference_class(PATH_LEGACY, "SynchAndChanEst")
    out = {}
    for i, (tag, par, n_sym, lead, fading) in enumerate(CASES):
        iq, bits = make_input(par, n_sym, lead, fading, 500 + i)
        blk = cls(par[0], par[1], par[2], par[3], list(par[4]), par[5], par[6], "/tmp/", "x", 0, 0)
        blk.diagnostic = 0            # the attribute :194 reads and the constructor forgets
        leg = cls_legacy(par[0], par[1], par[2], par[3], list(par[4]), par[5], par[6], "/tmp/", "x", 0)   # nothing patched
        same = True
        out[tag + "_iq"] = iq
        out[tag + "_bits"] = bits.astype(np.uint8)
        out[tag + "_par"] = np.array([par[0], par[1], par[2], par[3], par[4][0], par[4][1], par[5], par[6]], dtype=np.float64)
        for call in (1, 2):
            o = np.zeros(len(iq), np.complex64)
            with contextlib.redirect_stdout(io.StringIO()):          # :226 prints every data pointer
                blk.work([iq], [o])
            n_found = int(np.count_nonzero(blk.time_synch_ref[:, 2]))
            print(tag, "call", call, "syncs", n_found, "first", blk.time_synch_ref[:3, :2].tolist())
            k = "%s_call%d_" % (tag, call)
            out[k + "tsr"] = blk.time_synch_ref.copy()
            out[k + "H"] = blk.est_chan_freq_P.copy()
            out[k + "htime"] = blk.est_chan_time.copy()
            out[k + "esf"] = blk.est_synch_freq.copy()
            out[k + "edf"] = blk.est_data_freq.copy()
            out[k + "out"] = o
            o2 = np.zeros(len(iq), np.complex64)
            with contextlib.redirect_stdout(io.StringIO()):
                leg.work([iq], [o2])
            same = same and all(np.array_equal(a, b) for a, b in (
                (leg.time_synch_ref, blk.time_synch_ref), (leg.est_chan_freq_P, blk.est_chan_freq_P),
                (leg.est_chan_time, blk.est_chan_time), (leg.est_synch_freq, blk.est_synch_freq),
                (leg.est_data_freq, blk.est_data_freq), (o2, o)))
        print(tag, "unmodified LEGACY OFDMReceiver.SynchAndChanEst bit-identical:", same)
        out[tag + "_legacy_identical"] = np.array([int(same)])
    np.savez_compressed(os.path.join(G.HERE, "ref_rxofdm_table.npz"), **out)


if __name__ == "__main__":
    main()
