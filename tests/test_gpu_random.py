"""Randomised GPU parity (hypothesis): frame batches with random numerology, offsets, gains and channel taps vs the fp64 oracle."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from conftest import relerr
from oracle import ofdm_oracle as orc

pytestmark = pytest.mark.gpu


def _same_rows(got, ref, tol=2e-5):
    """Rows of `ref` that are finite must match within tol; rows the reference turned into NaN/inf (0 * inf on all-zero windows)
    must be non-finite here too."""
    got, ref = np.asarray(got), np.asarray(ref)
    got, ref = got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])
    fin = np.isfinite(ref).all(axis=1)
    assert np.array_equal(np.isfinite(got).all(axis=1), fin)
    if fin.any():
        assert relerr(got[fin], ref[fin]) < tol


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(logn=st.integers(6, 11), kd_frac=st.floats(0.3, 0.95), cp_frac=st.floats(0.04, 0.24), n_pat=st.integers(1, 3),
       n_frames=st.integers(1, 5), lead=st.integers(0, 7), gain=st.floats(0.05, 20.0), seed=st.integers(0, 2 ** 31 - 1),
       mod=st.sampled_from(["QPSK", "16QAM", "64QAM"]))
def test_batch_random_numerology(logn, kd_frac, cp_frac, n_pat, n_frames, lead, gain, seed, mod):
    import ofdm_mi355x as om
    N = 1 << logn
    cp = max(8, int(N * cp_frac))
    Kd = max(8, int((N - 2) * kd_frac) // 4 * 4)
    n_sym = 4 * n_pat
    L = N + cp
    rng = np.random.default_rng(seed)
    bps = orc.BITS_PER_SYMBOL[mod]
    taps = (rng.standard_normal(3) + 1j * rng.standard_normal(3)) * np.array([1.0, 0.4, 0.15])
    lead = min(lead, cp // 2)
    frame_len = n_sym * L + lead + 3
    iq = np.zeros((n_frames, frame_len), np.complex64)
    bits = rng.integers(0, 2, (n_frames, n_pat * 3 * Kd * bps)).astype(np.uint8)
    for f in range(n_frames):
        tx = orc.tx_modulate(bits[f], N, cp, N - 2, Kd, n_sym, modulation=mod)
        y = gain * orc.channel_apply(tx, taps, N)[:n_sym * L + 3]
        iq[f, lead:] = y
    rx = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.5, modulation=mod)
    nds = rx.data_symbols_per_frame(frame_len)
    d_iq = om.DeviceBuffer(iq.nbytes).upload(iq)
    d_eq = om.DeviceBuffer(max(8, n_frames * nds * Kd * 8))
    d_bu = om.DeviceBuffer(max(8, n_frames * nds * Kd * bps))
    d_tsr = om.DeviceBuffer(n_frames * 16)
    assert rx.demod_frames(d_iq, n_frames, frame_len, frame_len, d_eq, d_bu, om.BITS_UNPACKED, d_tsr) == nds == 3 * n_pat
    eq = d_eq.download(np.complex64, n_frames * nds * Kd).reshape(n_frames, nds, Kd)
    bu = d_bu.download(np.uint8, n_frames * nds * Kd * bps).reshape(n_frames, -1)
    tsr = d_tsr.download(np.int32, n_frames * 4).reshape(n_frames, 4)
    rows = [r for r in range(n_sym) if r % 4 != 3]
    for f in range(n_frames):
        o = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, 100, 0.5, force_fp64=True)
        o.work(iq[f], np.zeros(frame_len, np.complex64))
        # a detection that sits within fp32 rounding of the gate may legitimately differ: skip such frames
        m = np.max(np.abs(o.del_mat)) if o.del_mat is not None else 0.0
        if abs(m - 0.5 * o.MM) < 1e-3 * o.MM:
            continue
        assert tsr[f, 0] == o.time_synch_ref[0] and tsr[f, 1] == o.time_synch_ref[1]
        assert relerr(eq[f], o.est_data_freq[rows]) < 2e-5
        assert np.array_equal(bu[f], orc.demap_hard(eq[f].ravel(), mod))


@settings(max_examples=20, deadline=None, suppress_health_check=list(HealthCheck))
@given(case=st.integers(0, 9), dsss=st.booleans(), lead=st.integers(0, 40), cfo_hz=st.floats(-600.0, 600.0), fading=st.booleans(),
       n_fo=st.integers(1, 4), seed=st.integers(0, 2 ** 31 - 1), calls=st.integers(1, 3))
def test_sync_table_receivers_random(case, dsss, lead, cfo_hz, fading, n_fo, seed, calls):
    """The legacy sync-table receivers (SynchEstAndFO / SynchEstFOAndDSSS, every numerology case, random leads, carrier offsets,
    channels, candidate counts and call counts) vs their fp64 oracles.  Trials whose correlation peak sits within fp32 rounding of
    the 0.4*MM gate may legitimately differ: such draws are skipped; ill-conditioned equaliser bins (|H| tiny, SNR = 1e8) too."""
    import OFDMReceiver
    rng = np.random.default_rng(seed)
    cases = orc.DSSS_CASES if dsss else orc.FO_CASES
    n_symb, fs, N, sd, Kd = cases[case][:5]
    S, D = sd
    cp = N // 4
    n_data = sum(1 for s in range(n_symb) if s % (S + D) >= S)
    bits = rng.integers(0, 2, n_data * Kd * 2)
    tx = orc.tx_modulate(bits, N, cp, N - 2, Kd, n_symb, synch_dat=(S, D), zc_root=37, zc_segments=True, zc_parity_of_bins=True)
    if fading:
        tx = orc.channel_apply(tx, orc.REF_TAPS, N)[:len(tx) + 8]
    rx = tx * np.exp(1j * 2 * np.pi * cfo_hz / fs * np.arange(len(tx)))
    iq = np.concatenate([np.zeros(lead), rx, np.zeros(2 * cp)]).astype(np.complex64)
    fo_range = [float(x) for x in np.linspace(-1000.0, 1000.0, n_fo)] if n_fo > 1 else [0.0]
    o = (orc.FoDsssOracle if dsss else orc.FoOracle)(case, fo_range)
    blk = (OFDMReceiver.SynchEstFOAndDSSS if dsss else OFDMReceiver.SynchEstAndFO)(case, fo_range, "/tmp/ofdm_r_", "c", 0)
    # gate margins of every trial (oracle side): |peak - 0.4*MM| must not be within rounding
    for _ in range(calls):
        ro, rb = np.zeros(len(iq), np.complex64), np.zeros(len(iq), np.complex64)
        o.work(iq, ro)
        blk.work([iq], [rb])
        t = o.time_synch_ref
        n_sync = int(np.count_nonzero(t[:, 2])) if t[:, 2].any() else 0
        if np.any(np.abs(t[:n_sync, 2] - 0.4 * o.MM) < 2.0):
            return                                                  # a sync decided within rounding of the gate
        assert np.array_equal(blk.time_synch_ref[:, 0:2], t[:, 0:2])       # (0 differences in 159 measured call checks)
        if n_sync and np.abs(o.est_chan_freq_P[:n_sync][:, o.bins_used_P]).min() < 0.03:
            return
        assert relerr(blk.est_chan_freq_P, o.est_chan_freq_P) < 2e-5
        assert relerr(blk.est_data_freq, o.est_data_freq) < 2e-5
        if dsss:
            assert relerr(blk.est_data_freq_d, o.est_data_freq_d) < 2e-5
        if ro.any():
            assert relerr(rb, ro) < 2e-5


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(rows=st.integers(1, 24), n_sym_tx=st.integers(0, 20), lead=st.integers(0, 130), cut=st.integers(0, 90), calls=st.integers(1, 3),
       sigma=st.sampled_from([0.0, 0.01, 0.1]), seed=st.integers(0, 2 ** 31 - 1), pat=st.sampled_from([(1, 3), (1, 1), (2, 2)]))
def test_stream_block_differential_fuzz(rows, n_sym_tx, lead, cut, calls, sigma, seed, pat):
    """utsa_ofdm.SynchAndChanEst.work vs the fp64 oracle on buffers of arbitrary length (empty, shorter than a symbol, ragged tails),
    row counts that do or do not match the reference's reshape, several [S, D] patterns and call sequences: either both raise the
    same exception type (IndexError / ValueError, like NumPy does in the reference) or state and outputs agree."""
    import utsa_ofdm
    N, cp, Kd = 64, 16, 60
    S, D = pat
    rng = np.random.default_rng(seed)
    n_data = sum(1 for s in range(n_sym_tx) if s % (S + D) >= S)
    bits = rng.integers(0, 2, max(n_data, 1) * Kd * 2)
    tx = orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym_tx, synch_dat=(S, D), zc_segments=True) if n_sym_tx else np.zeros(0, complex)
    body = np.concatenate([np.zeros(lead), tx, np.zeros(30)])
    body = body[:max(0, len(body) - cut)]
    iq = (body + sigma * (rng.standard_normal(len(body)) + 1j * rng.standard_normal(len(body)))).astype(np.complex64)
    o = orc.RxOracle(rows, N, cp, N - 2, [S, D], Kd, 100, 0.7, force_fp64=True)
    blk = utsa_ofdm.SynchAndChanEst(rows, N, cp, N - 2, [S, D], Kd, 100, 0.7, "/tmp/ofdm_fz_", "c", 0, 0, "Fading")
    for _ in range(calls):
        ro, rb = np.zeros(len(iq), np.complex64), np.zeros(len(iq), np.complex64)
        eo = eb = None
        try:
            o.work(iq, ro)
        except (IndexError, ValueError) as e:
            eo = type(e)
        try:
            blk.work([iq], [rb])
        except (IndexError, ValueError) as e:
            eb = type(e)
        m = np.max(np.abs(o.del_mat)) if getattr(o, "del_mat", None) is not None and np.size(o.del_mat) else 0.0
        if abs(m - 0.7 * o.MM) < 2e-3 * o.MM:
            return                                          # detection decided within fp32 rounding of the gate
        assert eo == eb, (eo, eb)
        assert np.array_equal(np.asarray(blk.time_synch_ref)[0:2], np.asarray(o.time_synch_ref)[0:2])
        assert blk.count == o.count and blk.corr_obs == o.corr_obs
        if np.abs(o.est_chan_freq_P[0][o.bins_used_P]).min() > 1e-2:
            _same_rows(blk.est_data_freq, o.est_data_freq)
            if eo is None and np.isfinite(ro).all():
                assert relerr(rb, ro) < 2e-5 or not ro.any()
        if eo is not None:
            return


@settings(max_examples=30, deadline=None, suppress_health_check=list(HealthCheck))
@given(n_sym=st.integers(0, 56), lead=st.integers(0, 200), cut=st.integers(0, 120), tail=st.integers(0, 60), fading=st.booleans(),
       sigma=st.sampled_from([0.0, 0.02, 0.2]), gain=st.floats(0.1, 8.0), calls=st.integers(1, 3), seed=st.integers(0, 2 ** 31 - 1))
def test_tracker_block_differential_fuzz(n_sym, lead, cut, tail, fading, sigma, gain, calls, seed):
    """OFDMReceiver.SynchronizeAndEstimate(0) vs the fp64 oracle on arbitrary buffers (empty, truncated mid-symbol, more sync
    symbols than the block has rows, noise only) and call sequences: same exception type or same pointers / lags / estimates /
    data rows.  Draws with a correlation peak within rounding of the 0.5*MM acquisition gate are skipped."""
    import warnings
    import OFDMReceiver
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(seed)
    n_data = sum(1 for s in range(n_sym) if s % 4 >= 1)
    bits = rng.integers(0, 2, max(n_data, 1) * 120)
    tx = orc.tx_modulate(bits, 64, 16, 62, 60, n_sym, synch_dat=(1, 3), zc_root=23) if n_sym else np.zeros(0, complex)
    if fading and n_sym:
        tx = orc.channel_apply(tx, orc.REF_TAPS, 64)[:len(tx) + 8]
    body = np.concatenate([np.zeros(lead), gain * tx, np.zeros(tail)])
    body = body[:max(0, len(body) - cut)]
    iq = (body + sigma * (rng.standard_normal(len(body)) + 1j * rng.standard_normal(len(body)))).astype(np.complex64)
    o = orc.TrackerOracle(0)
    o.force_fp64 = True
    blk = OFDMReceiver.SynchronizeAndEstimate(0)
    for _ in range(calls):
        ro, rb = np.zeros(max(len(iq), 60), np.complex64), np.zeros(max(len(iq), 60), np.complex64)
        eo = eb = None
        try:
            o.work(iq, ro)
        except (IndexError, ValueError) as e:
            eo = type(e)
        try:
            blk.work([iq], [rb])
        except (IndexError, ValueError) as e:
            eb = type(e)
        t = o.time_synch_ref[0]
        n_sync = o.corr_obs + 1
        if n_sync > 0 and abs(t[0, 2] - 0.5 * o.MM) < 0.05:
            return
        assert eo == eb, (eo, eb)
        assert blk.corr_obs == o.corr_obs
        assert np.array_equal(blk.time_synch_ref[:, :, 0:2], o.time_synch_ref[:, :, 0:2])
        assert relerr(blk.time_synch_ref[:, :, 2], o.time_synch_ref[:, :, 2]) < 2e-5
        if np.isfinite(o.est_chan_freq_p).all():
            assert relerr(blk.est_chan_freq_p, o.est_chan_freq_p) < 2e-5
        if n_sync == 0 or np.abs(o.est_chan_freq_p[0, :min(n_sync, 12)][:, o.used_bins_data]).min() > 1e-2:
            _same_rows(blk.est_data_freq, o.est_data_freq)
            _same_rows(rb[None, :60], ro[None, :60])
        if eo is not None:
            return


@settings(max_examples=30, deadline=None, suppress_health_check=list(HealthCheck))
@given(logn=st.integers(6, 9), S=st.integers(1, 3), D=st.integers(1, 3), kd_frac=st.floats(0.2, 0.95), rows=st.integers(2, 30),
       n_pat=st.integers(0, 6), lead=st.integers(0, 90), cut=st.integers(0, 70), snr=st.sampled_from([50.0, 1e3, 1e8]),
       fading=st.booleans(), calls=st.integers(1, 3), seed=st.integers(0, 2 ** 31 - 1))
def test_table_mode_differential_fuzz(logn, S, D, kd_frac, rows, n_pat, lead, cut, snr, fading, calls, seed):
    """RXOFDM.synch_and_chan_est(table_mode=True) (== the legacy OFDMReceiver.SynchAndChanEst) with arbitrary constructor arguments
    vs the oracle: sync table, estimates, one data symbol per sync, output rows; same exception type when the reference raises
    (101st sync, reshape / output size)."""
    import RXOFDM
    N = 1 << logn
    cp = N // 4
    Ks = N - 2
    Kd = max(4, int((N - 2) * kd_frac) // 2 * 2)
    rng = np.random.default_rng(seed)
    n_sym = n_pat * (S + D)
    n_data = n_pat * D
    bits = rng.integers(0, 2, max(n_data, 1) * Kd * 2)
    tx = (orc.tx_modulate(bits, N, cp, Ks, Kd, n_sym, synch_dat=(S, D), zc_root=37, zc_segments=True, zc_parity_of_bins=True)
          if n_sym else np.zeros(0, complex))
    if fading and n_sym:
        tx = orc.channel_apply(tx, orc.REF_TAPS, N)[:len(tx) + 8]
    body = np.concatenate([np.zeros(lead), tx, np.zeros(2 * cp)])
    iq = body[:max(0, len(body) - cut)].astype(np.complex64)
    o = orc.FoOracle.from_params(rows, N, cp, Ks, (S, D), Kd, snr, force_fp64=True)
    blk = RXOFDM.synch_and_chan_est(rows, N, cp, Ks, [S, D], Kd, snr, "/tmp/ofdm_tf_", "c", 0, 0, table_mode=True)
    n_out = max(len(iq), (rows // (S + D)) * Kd)
    for _ in range(calls):
        ro, rb = np.zeros(n_out, np.complex64), np.zeros(n_out, np.complex64)
        eo = eb = None
        try:
            o.work(iq, ro)
        except (IndexError, ValueError) as e:
            eo = type(e)
        try:
            blk.work([iq], [rb])
        except (IndexError, ValueError) as e:
            eb = type(e)
        t = o.time_synch_ref
        n_sync = int(np.count_nonzero(t[:, 2]))
        if np.any(np.abs(t[:n_sync, 2] - 0.4 * o.MM) < 2.0):
            return
        assert eo == eb, (eo, eb)
        assert np.array_equal(blk.time_synch_ref[:, 0:2], t[:, 0:2])
        if n_sync and np.abs(o.est_chan_freq_P[:n_sync][:, o.bins_used_P]).min() < 0.03:
            return
        _same_rows(blk.est_chan_freq_P, o.est_chan_freq_P)
        _same_rows(blk.est_data_freq, o.est_data_freq)
        if eo is None and np.isfinite(ro).all():
            assert relerr(rb, ro) < 2e-5 or not ro.any()
        if eo is not None:
            return
