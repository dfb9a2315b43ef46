#!/usr/bin/env python3
"""Second reference data fixture for the hot path, build container only:

    python tests/golden/gen_golden_txdata0.py      -> tests/golden/ref_rx_txdata0.npz

LEGACY/gr-ofdm-tx/python/tx_data_0.pckl (1 x 3840 complex128 = 48 symbols x 80 samples, [1,3] pattern, ZC root 23, N = 64) is
read with the non-executing ndarray-pickle parser and pushed through the reference gr-utsa_ofdm SynchAndChanEst (two calls),
exactly like ref_rx_fixture64.npz was made from the TEST fixtures.  Only arrays are saved.
"""
import os

import numpy as np

import gen_golden as GG

FIXTURE = GG.G + "LEGACY/gr-ofdm-tx/python/tx_data_0.pckl"


def main():
    x = GG.load_ndarray(FIXTURE)
    assert x.shape == (1, 3840)
    iq = x[0].astype(np.complex64)
    outs, st = GG.ref_rx_run(iq, 48, 64, 16, 60, snr=100, gate=0.7, calls=2)
    res = {"iq": iq}
    for c, (o, s) in enumerate(zip(outs, st), start=1):
        print("call", c, "tsr", s["tsr"])
        for k, v in s.items():
            res["call%d_%s" % (c, k)] = np.asarray(v)
        res["call%d_out" % c] = o
    np.savez_compressed(os.path.join(GG.HERE, "ref_rx_txdata0.npz"), **res)


if __name__ == "__main__":
    main()
