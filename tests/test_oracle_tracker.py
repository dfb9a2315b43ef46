"""Pins oracle.TrackerOracle (regression-tracking receiver) to recorded runs of the reference
SynchronizeAndEstimate.py (tests/golden/gen_golden_tracker.py)."""
import warnings

import numpy as np
import pytest

from conftest import relerr
from oracle import ofdm_oracle as orc


@pytest.mark.parametrize("tag", ["clean", "fade", "noisy", "mimo_cfg", "txdata0"])
def test_tracker_oracle_matches_reference_runs(golden, tag):
    g = golden("ref_tracker.npz")
    case = int(g[tag + "_case"][0])
    o = orc.TrackerOracle(case)
    iq = g[tag + "_iq"]
    warnings.simplefilter("ignore")
    for call in (1, 2):
        out = np.zeros(len(iq), np.complex64)
        assert o.work(iq, out) == len(iq)
        k = "%s_call%d_" % (tag, call)
        assert o.corr_obs == int(g[k + "corr_obs"][0])
        assert np.array_equal(o.time_synch_ref[:, :, 0:2], g[k + "tsr"][:, :, 0:2])
        assert relerr(o.time_synch_ref[:, :, 2], g[k + "tsr"][:, :, 2]) < 1e-12
        assert relerr(o.est_chan_freq_p, g[k + "Hp"]) < 1e-12
        assert relerr(o.est_chan_freq_n, g[k + "Hn"]) < 1e-12
        assert relerr(o.est_chan_impulse, g[k + "imp"]) < 1e-12
        assert relerr(o.est_synch_freq, g[k + "esf"]) < 1e-12
        if case == 0:
            assert relerr(o.est_data_freq, g[k + "edf"]) < 1e-11
            assert relerr(out, g[k + "out"]) < 1e-6
        else:
            assert not out.any() and not g[k + "out"].any()
    if tag == "txdata0":
        # the reference's own fixture: 12 syncs, every peak = MM = 62 (a noiseless ZC symbol), 320 samples apart until the
        # regression takes over
        t = g["txdata0_call1_tsr"][0]
        assert int(g["txdata0_call1_corr_obs"][0]) == 11 and np.allclose(t[:12, 2], 62.0, atol=1e-6)
        assert np.array_equal(t[:6, 0], 11 + 320 * np.arange(6))
    elif case == 0:
        # call 1 found every pattern: its rows de-map to the transmitted bits (checked on the recorded reference rows)
        edf = g[tag + "_call1_edf"][0]
        n_sync = int(g[tag + "_call1_corr_obs"][0]) + 1
        got = orc.demap_hard(edf[:n_sync * 3].ravel(), "QPSK")
        assert np.count_nonzero(got != g[tag + "_bits"][:n_sync * 3 * 120]) == 0
