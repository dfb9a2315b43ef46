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


@pytest.mark.parametrize("tag", ["d1", "d3", "d4", "d8", "d10"])
def test_dsss_oracle_matches_reference_runs(golden, tag):
    """oracle.FoDsssOracle vs recorded runs of the reference SynchEstFOAndDSSS.py (tests/golden/gen_golden_dsss.py)."""
    g = golden("ref_dsss.npz")
    case = int(g[tag + "_meta"][0])
    o = orc.FoDsssOracle(case, list(g[tag + "_fo_range"]))
    iq = g[tag + "_iq"]
    for call in (1, 2):
        out = np.zeros(len(iq), np.complex64)
        assert o.work(iq, out) == len(iq)
        k = "%s_call%d_" % (tag, call)
        assert np.array_equal(o.time_synch_ref, g[k + "tsr"])
        assert o.dmax_tmp_ind == int(g[k + "fo_idx"][0])
        assert relerr(o.est_chan_freq_P, g[k + "H"]) < 1e-12
        assert relerr(o.est_data_freq, g[k + "edf"]) < 1e-11
        assert relerr(o.est_data_freq_d, g[k + "edfd"]) < 1e-11
        assert out.any() and relerr(out, g[k + "out"]) < 1e-6            # written on every call (no count gate)
    # the despread symbols carry the transmitted QPSK bits (small residual carrier offset, common phase absorbed per sync)
    n_sync = int(np.count_nonzero(o.time_synch_ref[:, 2]))
    n_spread = o.est_data_freq_d.shape[1]
    got = orc.demap_hard(o.est_data_freq_d[:n_sync].ravel(), "QPSK")
    ref = g[tag + "_bits"][:n_sync * n_spread * 2]
    assert np.count_nonzero(got != ref) == 0


def test_dsss_oracle_unbound_local_on_short_second_buffer():
    """DS:392 uses the loop's locals outside the guard: row 0 of an earlier call against a buffer too short for its data
    symbol raises UnboundLocalError in the reference; the oracle keeps that."""
    o = orc.FoDsssOracle(1, [0.0])
    n_symb, fs, N, sd, Kd, dsss = orc.DSSS_CASES[1]
    rng = np.random.default_rng(0)
    sym = orc.map_bits(rng.integers(0, 2, 12 * (Kd // dsss) * 2), "QPSK").reshape(12, Kd // dsss)
    iq = orc.tx_modulate(None, N, N // 4, N - 2, Kd, n_symb, synch_dat=sd, zc_root=37, zc_segments=True,
                         zc_parity_of_bins=True, data_symbols=orc.dsss_spread(sym, dsss, Kd)).astype(np.complex64)
    o.work(iq, np.zeros(len(iq), np.complex64))
    short = iq[:int(o.time_synch_ref[0][0]) + sd[0] * (N + N // 4) + N - 2]
    with pytest.raises(UnboundLocalError):
        o.work(short, np.zeros(len(short), np.complex64))


@pytest.mark.parametrize("tag", ["chain", "s2", "n256"])
def test_table_mode_oracle_matches_gr_rxofdm_runs(golden, tag):
    """FoOracle.from_params (no carrier-offset search) vs recorded runs of gr-RXOFDM's synch_and_chan_est.work
    (tests/golden/gen_golden_rxofdm_table.py; the instance is given the `diagnostic` attribute its constructor forgets)."""
    g = golden("ref_rxofdm_table.npz")
    assert int(g[tag + "_legacy_identical"][0]) == 1      # same arrays from the unmodified LEGACY OFDMReceiver.SynchAndChanEst
    p = g[tag + "_par"]
    o = orc.FoOracle.from_params(int(p[0]), int(p[1]), int(p[2]), int(p[3]), (int(p[4]), int(p[5])), int(p[6]), float(p[7]))
    iq = g[tag + "_iq"]
    for call in (1, 2):
        out = np.zeros(len(iq), np.complex64)
        assert o.work(iq, out) == len(iq)
        k = "%s_call%d_" % (tag, call)
        assert np.array_equal(o.time_synch_ref, g[k + "tsr"])
        assert relerr(o.est_chan_freq_P, g[k + "H"]) < 1e-12
        assert relerr(o.est_chan_time, g[k + "htime"]) < 1e-12
        assert relerr(o.est_synch_freq, g[k + "esf"]) < 1e-12
        assert relerr(o.est_data_freq, g[k + "edf"]) < 1e-11
        assert relerr(out, g[k + "out"]) < 1e-6 or not g[k + "out"].any()
