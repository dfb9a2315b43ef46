"""Pins oracle.FoOracle (CFO-search receiver) to recorded runs of the reference SynchEstAndFO.py."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ofdm_oracle as orc


@pytest.mark.parametrize("tag", ["c0", "c3", "c6", "c9"])
def test_fo_oracle_matches_reference_runs(golden, tag):
    g = golden("ref_fo.npz")
    case = int(g[tag + "_meta"][0])
    o = orc.FoOracle(case, list(g[tag + "_fo_range"]))
    iq = g[tag + "_iq"]
    for call in (1, 2):
        out = np.zeros(len(iq), np.complex64)
        assert o.work(iq, out) == len(iq)
        assert np.array_equal(o.time_synch_ref, g["%s_call%d_tsr" % (tag, call)])
        assert o.dmax_tmp_ind == int(g["%s_call%d_fo_idx" % (tag, call)][0])
        assert relerr(o.est_chan_freq_P, g["%s_call%d_H" % (tag, call)]) < 1e-12
        assert relerr(o.est_chan_time, g["%s_call%d_htime" % (tag, call)]) < 1e-12
        assert relerr(o.est_synch_freq, g["%s_call%d_esf" % (tag, call)]) < 1e-12
        assert relerr(o.est_data_freq, g["%s_call%d_edf" % (tag, call)]) < 1e-11
        assert relerr(out, g["%s_call%d_out" % (tag, call)]) < 1e-6 or not g["%s_call%d_out" % (tag, call)].any()
    assert np.count_nonzero(o.time_synch_ref[:, 2]) >= 3          # several syncs per buffer were found
