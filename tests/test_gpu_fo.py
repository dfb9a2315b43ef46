"""GPU parity of the CFO-search receiver: OFDMReceiver.SynchEstAndFO through the C ABI vs recorded runs of
the reference block (tests/golden/ref_fo.npz) and vs oracle.FoOracle on seeded inputs.

Tolerances as in test_gpu_parity.py: time_synch_ref[:, 0:2] exact, [:, 2] = int(max|corr|) within +-1 (fp32 vs fp64
truncation), fp32 FFT / estimate / equaliser outputs 1e-5 norm-relative."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ofdm_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _block(case, fo_range, **kw):
    import OFDMReceiver
    return OFDMReceiver.SynchEstAndFO(case, list(fo_range), "/tmp/ofdm_fo_", "cest", 0, **kw)


def _check_table(got, ref):
    assert np.array_equal(got[:, 0:2], ref[:, 0:2])
    assert np.max(np.abs(got[:, 2] - ref[:, 2])) <= 1


def _make_input(case, cfo_hz, lead, fading, seed, n_sym=None, tail=None):
    n_symb, fs, N, sd, Kd = orc.FO_CASES[case]
    S, D = sd
    cp = N // 4
    rng = np.random.default_rng(seed)
    n_sym = n_symb if n_sym is None else n_sym
    n_data = sum(1 for s in range(n_sym) if s % (S + D) >= S)
    bits = rng.integers(0, 2, n_data * Kd * 2)
    tx = orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym, synch_dat=(S, D), zc_root=37, zc_segments=True, zc_parity_of_bins=True)
    if fading:
        tx = orc.channel_apply(tx, orc.REF_TAPS, N)[:len(tx) + 8]
    rx = tx * np.exp(1j * 2 * np.pi * cfo_hz / fs * np.arange(len(tx)))
    return np.concatenate([np.zeros(lead), rx, np.zeros(2 * cp if tail is None else tail)]).astype(np.complex64), bits


@pytest.mark.parametrize("tag", ["c0", "c3", "c6", "c9"])
def test_fo_block_on_reference_runs(golden, tag):
    g = golden("ref_fo.npz")
    blk = _block(int(g[tag + "_meta"][0]), g[tag + "_fo_range"])
    iq = g[tag + "_iq"]
    for call in (1, 2):
        out = np.zeros(len(iq), np.complex64)
        assert blk.work([iq], [out]) == len(iq)
        k = "%s_call%d_" % (tag, call)
        _check_table(blk.time_synch_ref, g[k + "tsr"])
        assert blk.dmax_tmp_ind == int(g[k + "fo_idx"][0])
        assert relerr(blk.est_chan_freq_P, g[k + "H"]) < TOL
        assert relerr(blk.est_chan_time, g[k + "htime"]) < TOL
        assert relerr(blk.est_synch_freq, g[k + "esf"]) < TOL
        assert relerr(blk.est_data_freq, g[k + "edf"]) < TOL
        if call == 1:
            assert not out.any()                                   # count == 0: nothing emitted (:366)
        else:
            assert relerr(out, g[k + "out"]) < TOL
        assert blk.count == call and blk.cor_obs == 0


@pytest.mark.parametrize("case,fo_range,cfo_hz,fading,seed", [
    (1, [-22000, -9000, 0, 9000, 22000], 9000.0, False, 22),
    (5, [0, 20000, 41000], -41000.0, True, 23),
    (7, [-30000, 0, 30000, 61000], -30000.0, True, 27),
    (8, [-57000, 0], 57000.0, False, 21),
])
def test_fo_block_true_rotators_vs_oracle(case, fo_range, cfo_hz, fading, seed):
    """py2_rotators=False: the candidates really differ, the search must pick the transmitted offset; the LS estimate still
    uses the LAST candidate's sync vector and the data the LAST trial's pick (FO:268-274,283,339).  Oracle-only parity
    (the reference file cannot produce these rotators under its own Python-2 semantics).
    The offsets are chosen so that (offset + last candidate) is not a whole number of bins: a whole-bin shift leaves exact
    zeros in the LS estimate, and with SNR = 1e8 the equaliser gain conj(H)/(1e-8+|H|^2) turns rounding noise at such a bin
    into O(1) output -- ill-conditioned in the reference itself, so no tolerance is meaningful there (asserted below)."""
    iq, _ = _make_input(case, cfo_hz, 4, fading, seed)
    o = orc.FoOracle(case, fo_range, py2_rotators=False)
    blk = _block(case, fo_range, py2_rotators=False)
    for call in (1, 2):
        ro, rb = np.zeros(len(iq), np.complex64), np.zeros(len(iq), np.complex64)
        o.work(iq, ro)
        blk.work([iq], [rb])
        _check_table(blk.time_synch_ref, o.time_synch_ref)
        assert blk.dmax_tmp_ind == o.dmax_tmp_ind
        n_sync = int(np.count_nonzero(o.time_synch_ref[:, 2]))
        assert np.abs(o.est_chan_freq_P[:n_sync][:, o.bins_used_P]).min() > 0.03       # well-conditioned equaliser
        assert relerr(blk.est_chan_freq_P, o.est_chan_freq_P) < TOL
        assert relerr(blk.est_chan_time, o.est_chan_time) < TOL
        assert relerr(blk.est_synch_freq, o.est_synch_freq) < TOL
        assert relerr(blk.est_data_freq, o.est_data_freq) < TOL
        assert relerr(blk.eq_gain, o.eq_gain) < TOL
        assert relerr(rb, ro) < TOL or not ro.any()
    assert np.count_nonzero(o.time_synch_ref[:, 2]) >= 1


def test_fo_block_recovers_bits_with_matching_candidate():
    """Intended rotators, one candidate that cancels the transmitted offset (cfo[fo] multiplies by e^{+j 2 pi fo n/fs}, so the
    candidate is -offset): every sync's data symbol de-maps to the transmitted bits; without the rotation it does not.
    The offset is chosen so that the phase advance over one symbol period (L samples) is a multiple of 2 pi: the block has
    no common-phase tracking between the sync symbol and its data symbol (FO:335-358)."""
    case, cfo_hz = 2, 12000.0
    n_symb, fs, N, sd, Kd = orc.FO_CASES[case]
    assert (cfo_hz * (N + N // 4)) % fs == 0
    iq, bits = _make_input(case, cfo_hz, 0, False, 11)
    errs = {}
    for name, fo_range in (("matched", [-cfo_hz]), ("none", [0.0])):
        blk = _block(case, fo_range, py2_rotators=False)
        blk.work([iq], [np.zeros(len(iq), np.complex64)])
        n_sync = int(np.count_nonzero(blk.time_synch_ref[:, 2]))
        assert n_sync == n_symb // 2 and blk.dmax_tmp_ind == 0
        got = orc.demap_hard(blk.est_data_freq[:n_sync].ravel(), "QPSK")
        errs[name] = int(np.count_nonzero(got != bits[:n_sync * Kd * 2]))
    assert errs["matched"] == 0 and errs["none"] > 500


def test_fo_block_errors_follow_the_reference():
    case = 0
    n_symb, fs, N, sd, Kd = orc.FO_CASES[case]
    L = N + N // 4
    # a 101st sync in one buffer: IndexError (FO:294-296)
    iq, _ = _make_input(case, 0.0, 0, False, 3, n_sym=2 * 104)
    o = orc.FoOracle(case, [0.0])
    with pytest.raises(IndexError):
        o.work(iq, np.zeros(len(iq), np.complex64))
    blk = _block(case, [0.0])
    with pytest.raises(IndexError):
        blk.work([iq], [np.zeros(len(iq), np.complex64)])
    # row 0 of an earlier call against a later, shorter buffer whose data slice is one sample short: ValueError (FO:338-339)
    iq, _ = _make_input(case, 0.0, 0, False, 4)
    o = orc.FoOracle(case, [0.0])
    blk = _block(case, [0.0])
    o.work(iq, np.zeros(len(iq), np.complex64))
    blk.work([iq], [np.zeros(len(iq), np.complex64)])
    short = iq[:int(o.time_synch_ref[0][0]) + L + N - 1]
    with pytest.raises(ValueError):
        o.work(short, np.zeros(len(short), np.complex64))
    with pytest.raises(ValueError):
        blk.work([short], [np.zeros(len(short), np.complex64)])
    # out-of-range case: the reference prints and then dies on the first missing attribute
    with pytest.raises(AttributeError):
        _block(17, [0.0])


def test_fo_block_quiet_buffer_keeps_state():
    """No sync in the buffer: tables untouched, cor_obs reset to 0, row 0 (all zeros) 'demodulated' like the reference does
    from the second call on (FO:332 runs for P = 0 because cor_obs was reset to 0)."""
    case = 4
    rng = np.random.default_rng(5)
    iq = (1e-3 * (rng.standard_normal(2000) + 1j * rng.standard_normal(2000))).astype(np.complex64)
    o = orc.FoOracle(case, [0.0, 100.0])
    blk = _block(case, [0.0, 100.0])
    for _ in range(2):
        ro, rb = np.zeros(len(iq), np.complex64), np.zeros(len(iq), np.complex64)
        o.work(iq, ro)
        blk.work([iq], [rb])
        assert not blk.time_synch_ref.any() and not o.time_synch_ref.any()
        assert np.allclose(blk.est_data_freq, o.est_data_freq, atol=1e-6)
        assert np.allclose(rb, ro, atol=1e-6)


# ------------------------------------------------------------------------------------------ DSSS variant
def _dsss_block(case, fo_range, **kw):
    import OFDMReceiver
    return OFDMReceiver.SynchEstFOAndDSSS(case, list(fo_range), "/tmp/ofdm_fo_", "cest", 0, **kw)


@pytest.mark.parametrize("tag", ["d1", "d3", "d4", "d8", "d10"])
def test_dsss_block_on_reference_runs(golden, tag):
    """OFDMReceiver.SynchEstFOAndDSSS vs recorded runs of the reference block (tests/golden/ref_dsss.npz)."""
    g = golden("ref_dsss.npz")
    blk = _dsss_block(int(g[tag + "_meta"][0]), g[tag + "_fo_range"])
    iq = g[tag + "_iq"]
    for call in (1, 2):
        out = np.zeros(len(iq), np.complex64)
        assert blk.work([iq], [out]) == len(iq)
        k = "%s_call%d_" % (tag, call)
        _check_table(blk.time_synch_ref, g[k + "tsr"])
        assert blk.dmax_tmp_ind == int(g[k + "fo_idx"][0])
        assert relerr(blk.est_chan_freq_P, g[k + "H"]) < TOL
        assert relerr(blk.est_data_freq, g[k + "edf"]) < TOL
        assert relerr(blk.est_data_freq_d, g[k + "edfd"]) < TOL
        assert out.any() and relerr(out, g[k + "out"]) < TOL            # emitted on every call
        assert blk.count == call and blk.cor_obs == 0
    n_sync = int(np.count_nonzero(blk.time_synch_ref[:, 2]))
    n_spread = blk.est_data_freq_d.shape[1]
    got = orc.demap_hard(blk.est_data_freq_d[:n_sync].ravel(), "QPSK")
    assert np.array_equal(got, g[tag + "_bits"][:n_sync * n_spread * 2])


def test_dsss_block_unbound_local_like_the_reference():
    o = orc.FoDsssOracle(1, [0.0])
    blk = _dsss_block(1, [0.0])
    n_symb, fs, N, sd, Kd, dsss = orc.DSSS_CASES[1]
    rng = np.random.default_rng(0)
    sym = orc.map_bits(rng.integers(0, 2, 12 * (Kd // dsss) * 2), "QPSK").reshape(12, Kd // dsss)
    iq = orc.tx_modulate(None, N, N // 4, N - 2, Kd, n_symb, synch_dat=sd, zc_root=37, zc_segments=True,
                         zc_parity_of_bins=True, data_symbols=orc.dsss_spread(sym, dsss, Kd)).astype(np.complex64)
    ro, rb = np.zeros(len(iq), np.complex64), np.zeros(len(iq), np.complex64)
    o.work(iq, ro)
    blk.work([iq], [rb])
    assert relerr(rb, ro) < TOL
    short = iq[:int(o.time_synch_ref[0][0]) + sd[0] * (N + N // 4) + N - 2]
    with pytest.raises(UnboundLocalError):
        o.work(short, np.zeros(len(short), np.complex64))
    with pytest.raises(UnboundLocalError):
        blk.work([short], [np.zeros(len(short), np.complex64)])
    with pytest.raises(AttributeError):
        _dsss_block(11, [0.0])


# ------------------------------------------------------------------------------------------ gr-RXOFDM table mode
@pytest.mark.parametrize("tag", ["chain", "s2", "n256"])
def test_rxofdm_table_mode_on_reference_runs(golden, tag):
    """RXOFDM.synch_and_chan_est(..., table_mode=True) vs recorded runs of gr-RXOFDM's own work()
    (tests/golden/ref_rxofdm_table.npz; "chain" is the ofdm_chain.py:83 wiring with K = N sync bins)."""
    import RXOFDM
    g = golden("ref_rxofdm_table.npz")
    p = g[tag + "_par"]
    blk = RXOFDM.synch_and_chan_est(int(p[0]), int(p[1]), int(p[2]), int(p[3]), [int(p[4]), int(p[5])], int(p[6]), float(p[7]),
                                    "/tmp/", "x", 0, 0, table_mode=True)
    assert type(blk).__name__ == "synch_and_chan_est_table"
    # the unmodified LEGACY OFDMReceiver.SynchAndChanEst produced bit-identical arrays when the goldens were recorded
    assert int(g[tag + "_legacy_identical"][0]) == 1
    import OFDMReceiver
    leg = OFDMReceiver.SynchAndChanEst(int(p[0]), int(p[1]), int(p[2]), int(p[3]), [int(p[4]), int(p[5])], int(p[6]), float(p[7]),
                                       "/tmp/", "x", 0)
    iq = g[tag + "_iq"]
    for call in (1, 2):
        out = np.zeros(len(iq), np.complex64)
        assert blk.work([iq], [out]) == len(iq)
        out_l = np.zeros(len(iq), np.complex64)
        leg.work([iq], [out_l])
        assert np.array_equal(out_l, out) and np.array_equal(leg.est_data_freq, blk.est_data_freq)
        k = "%s_call%d_" % (tag, call)
        _check_table(blk.time_synch_ref, g[k + "tsr"])
        assert relerr(blk.est_chan_freq_P, g[k + "H"]) < TOL
        assert relerr(blk.est_chan_time, g[k + "htime"]) < TOL
        assert relerr(blk.est_synch_freq, g[k + "esf"]) < TOL
        assert relerr(blk.est_data_freq, g[k + "edf"]) < TOL
        if call == 1:
            assert not out.any()
        else:
            assert relerr(out, g[k + "out"]) < TOL
    # one data symbol per sync: de-maps to the first data symbol of every pattern
    S, D, Kd = int(p[4]), int(p[5]), int(p[6])
    n_sync = int(np.count_nonzero(blk.time_synch_ref[:, 2]))
    got = orc.demap_hard(blk.est_data_freq[:n_sync].ravel(), "QPSK").reshape(n_sync, Kd * 2)
    ref = g[tag + "_bits"].reshape(-1, Kd * 2)[0::D][:n_sync]
    assert np.array_equal(got, ref)


def test_rxofdm_table_mode_short_slice_is_zero_padded():
    """Without the rotator product a data slice that is one sample short is zero-padded by fft(x, N) (RXc:228-230) where
    SynchEstAndFO raises; same inputs as the FO error test, vs the oracle with the fp64 yardstick."""
    import RXOFDM
    par = (48, 64, 16, 62, (1, 1), 12, 1e8)
    iq, _ = _make_input(0, 0.0, 0, False, 4)
    o = orc.FoOracle.from_params(*par, force_fp64=True)
    blk = RXOFDM.synch_and_chan_est(par[0], par[1], par[2], par[3], list(par[4]), par[5], par[6], "/tmp/", "x", 0, 0,
                                    table_mode=True)
    o.work(iq, np.zeros(len(iq), np.complex64))
    blk.work([iq], [np.zeros(len(iq), np.complex64)])
    short = iq[:int(o.time_synch_ref[0][0]) + 80 + 64 - 1]
    ro, rb = np.zeros(len(iq), np.complex64), np.zeros(len(iq), np.complex64)      # room for the corr_size rows
    o.work(short, ro)
    blk.work([short], [rb])
    _check_table(blk.time_synch_ref, o.time_synch_ref)
    assert relerr(blk.est_data_freq, o.est_data_freq) < TOL
    assert ro.any() and relerr(rb, ro) < TOL
