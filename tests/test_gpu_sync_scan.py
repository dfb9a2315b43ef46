"""GPU parity of the batch path's sync search with frames that do NOT start at sample 0 (run with -m gpu).

Every frame is preceded by its own random lead of 0..L-1 noise samples, so the search of every frame ends at a different
trial, deep inside the frame or -- with a trial cap -- not at all.  `time_synch_ref` must equal the oracle's (position and lag
exactly, int(peak) within 1), the equalised symbols must match at 1e-5, and the screened search (anchor trials evaluated
exactly, the trials between them screened by the sliding recurrence) must take the same decisions and return the same arrays (to
the last bits: 2e-6 of the peak) as the exhaustive trial-by-trial search."""
import numpy as np
import pytest

from conftest import assert_close, poisoned, relerr
from oracle import ofdm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def om():
    import ofdm_mi355x
    ofdm_mi355x.load()
    return ofdm_mi355x


def _frames_with_leads(N, cp, Kd, n_sym, leads, sigma, seed, lead_kind="noise", taps=orc.REF_TAPS):
    rng = np.random.default_rng(seed)
    L = N + cp
    fl = n_sym * L + 7
    nb = (n_sym // 4) * 3 * Kd * 2
    iq = np.zeros((len(leads), fl), np.complex64)
    bits = rng.integers(0, 2, (len(leads), nb)).astype(np.uint8)
    for f, ld in enumerate(leads):
        tx = orc.channel_apply(orc.tx_modulate(bits[f], N, cp, N - 2, Kd, n_sym), taps, N)
        pre = np.zeros(ld, complex) if lead_kind == "zeros" else 0.3 * (rng.standard_normal(ld) + 1j * rng.standard_normal(ld))
        x = np.concatenate([pre, tx])[:fl]
        x = np.concatenate([x, np.zeros(fl - len(x))])
        iq[f] = (x + sigma * (rng.standard_normal(fl) + 1j * rng.standard_normal(fl))).astype(np.complex64)
    return bits, iq


def _run(om, rx, iq, Kd, exhaustive, max_trials=0):
    n_frames, fl = iq.shape
    assert rx.set_sync_search(exhaustive) == (not exhaustive)
    rx.set_max_trials(max_trials)
    nds = rx.data_symbols_per_frame(fl)
    d_iq = om.DeviceBuffer(iq.nbytes).upload(iq)
    d_eq = poisoned(om, n_frames * nds * Kd * 8)           # 0xFF: a row the kernels do not write cannot pass as zeros
    d_b = poisoned(om, n_frames * nds * Kd * 2)
    d_tsr = poisoned(om, n_frames * 16)
    rx.demod_frames(d_iq, n_frames, fl, fl, d_eq, d_b, om.BITS_UNPACKED, d_tsr)
    eq = d_eq.download(np.complex64, n_frames * nds * Kd).reshape(n_frames, nds, Kd)
    b = d_b.download(np.uint8, n_frames * nds * Kd * 2).reshape(n_frames, -1)
    tsr = d_tsr.download(np.int32, n_frames * 4).reshape(n_frames, 4)
    H = np.stack([rx.frame_state(f)["chan_freq"] for f in range(n_frames)])
    return eq, b, tsr, H


@pytest.mark.parametrize("N,cp,Kd,n_sym,n_frames,sigma,max_lead", [
    (64, 16, 60, 12, 40, 0.05, None), (128, 32, 100, 8, 16, 0.05, None), (256, 64, 180, 8, 12, 0.1, None),
    (512, 36, 300, 8, 8, 0.05, None), (1024, 72, 600, 8, 8, 0.05, None), (2048, 144, 1200, 8, 6, 0.05, None),
    (4096, 288, 2400, 4, 3, 0.02, 1500),
])
def test_random_leads_match_the_oracle(om, N, cp, Kd, n_sym, n_frames, sigma, max_lead):
    L = N + cp
    rng = np.random.default_rng(N)
    leads = rng.integers(0, max_lead or L, n_frames)
    leads[0] = 0                                               # one aligned frame among them
    leads[1] = (max_lead or L) - 1
    bits, iq = _frames_with_leads(N, cp, Kd, n_sym, leads, sigma, seed=N + 1)
    fl = iq.shape[1]
    rx = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 30, 0.7)
    eq_s, b_s, tsr_s, H_s = _run(om, rx, iq, Kd, exhaustive=False)
    eq_x, b_x, tsr_x, H_x = _run(om, rx, iq, Kd, exhaustive=True)
    # same decisions; the arrays agree to the last bits (two separately compiled kernels around the same trial function: fused
    # multiply-adds may be contracted differently, so "bit for bit" is not promised -- 2e-6 of the peak is)
    assert np.array_equal(tsr_s, tsr_x) and np.array_equal(b_s, b_x)
    fin = np.isfinite(eq_x)
    assert np.array_equal(fin, np.isfinite(eq_s))
    assert relerr(np.where(fin, eq_s, 0), np.where(fin, eq_x, 0)) < 2e-6 and relerr(H_s, H_x) < 2e-6
    rows = [r for r in range(n_sym) if r % 4 != 3]
    hits = set()
    for f in range(n_frames):
        o = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, 30, 0.7, force_fp64=True)
        o.work(iq[f], np.zeros(fl, np.complex64))
        assert tsr_s[f, 3] == 1 and tsr_s[f, 0] == o.time_synch_ref[0] and tsr_s[f, 1] == o.time_synch_ref[1], (f, leads[f], tsr_s[f], o.time_synch_ref)
        assert abs(tsr_s[f, 2] - o.time_synch_ref[2]) <= 1
        hits.add(int(tsr_s[f, 0]))
        nd = eq_s.shape[1]
        with np.errstate(all="ignore"):
            ref = o.est_data_freq[rows][:nd]
        # EVERY row: patterns the lead pushed past the frame end keep the zeros of :88 (guard :223), windows that run past the
        # end are zero-padded by np.fft.fft(x, N) (:230), empty ones divide 0 by 0 (:233) -- all of it must be reproduced
        nan_ref = ~np.isfinite(ref).all(axis=1)
        assert np.array_equal(nan_ref, ~np.isfinite(eq_s[f]).all(axis=1)), (f, leads[f])
        zero_ref = ~nan_ref & ~ref.any(axis=1)
        assert not eq_s[f][zero_ref].any(), (f, leads[f])
        live = ~nan_ref & ~zero_ref
        assert_close(eq_s[f][live], ref[live], "frame %d (lead %d)" % (f, leads[f]))
        assert b_s.max() <= 1
    assert len(hits) > min(4, n_frames - 2)                    # the search really ended at different trials


@pytest.mark.parametrize("N,cp,Kd,n_sym,n_frames,gate,taps", [
    (1024, 72, 600, 8, 120, 0.7, "ref"), (2048, 144, 1200, 8, 160, 0.7, "delay"), (2048, 144, 1200, 8, 96, 0.3, "rayleigh"),
    (2048, 144, 1200, 8, 96, 0.7, "rayleigh"), (4096, 288, 2400, 4, 64, 0.7, "delay"), (4096, 288, 2400, 4, 48, 0.3, "rayleigh")])
def test_dense_lead_sweep_screened_equals_exhaustive(om, N, cp, Kd, n_sym, n_frames, gate, taps):
    """The skip of cold stretches is decided trial by trial from the anchor's lag vector: how far it carries depends on where the
    sync lies relative to the anchor, on the channel and on the gate.  Frames with leads spread evenly over 0 ... 2.1 L (anchors
    at every phase of the approach, long skips, skips that end a few trials short, peaks inside the first block), a one-sample
    delay / the reference taps / an 8-tap Rayleigh draw, gate 0.7 and 0.3: decisions identical to the exhaustive search, arrays
    to 2e-6 (the exhaustive kernel is pinned to the oracle above)."""
    L = N + cp
    rng = np.random.default_rng(N + n_frames)
    leads = np.linspace(0, 2.1 * L, n_frames).astype(int) + rng.integers(0, 5, n_frames)
    leads[0] = 0
    if taps == "ref":
        h = orc.REF_TAPS
    elif taps == "delay":
        h = np.array([0.0, 1.0])
    else:
        h = (rng.standard_normal(8) + 1j * rng.standard_normal(8)) * np.exp(-0.4 * np.arange(8))
    n_sym_tx = n_sym + 4                                       # the late frames still hold whole patterns
    bits, iq = _frames_with_leads(N, cp, Kd, n_sym_tx, leads, 0.03, seed=N + 7, taps=h)
    rx = om.RxEngine(n_sym_tx, N, cp, N - 2, (1, 3), Kd, 30, gate)
    eq_s, b_s, tsr_s, H_s = _run(om, rx, iq, Kd, exhaustive=False)
    eq_x, b_x, tsr_x, H_x = _run(om, rx, iq, Kd, exhaustive=True)
    assert np.array_equal(tsr_s, tsr_x), np.nonzero((tsr_s != tsr_x).any(axis=1))[0]
    assert np.array_equal(b_s, b_x)
    fin = np.isfinite(eq_x)
    assert np.array_equal(fin, np.isfinite(eq_s))
    assert relerr(np.where(fin, eq_s, 0), np.where(fin, eq_x, 0)) < 2e-6 and relerr(H_s, H_x) < 2e-6
    assert tsr_s[:, 3].sum() >= n_frames - 2                   # (a Rayleigh draw may leave a frame below the gate in both searches)
    assert len(set(tsr_s[:, 0].tolist())) > n_frames // 2


def test_zero_leads_noise_only_frames_and_trial_cap(om):
    """Leads of exact zeros (the window energy is 0: the screen must hand those trials to the exact evaluation), a frame of pure
    noise (no sync anywhere), and a trial cap below some frames' sync position: identical to the exhaustive search and to the
    oracle's verdict."""
    N, cp, Kd, n_sym = 256, 64, 180, 8
    L = N + cp
    leads = np.array([0, 5, 100, 250, 319, 30, 310, 200])
    bits, iq = _frames_with_leads(N, cp, Kd, n_sym, leads, 0.0, seed=9, lead_kind="zeros")
    rng = np.random.default_rng(3)
    iq[5] = (0.2 * (rng.standard_normal(iq.shape[1]) + 1j * rng.standard_normal(iq.shape[1]))).astype(np.complex64)
    fl = iq.shape[1]
    rx = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7)
    for cap in (0, 150):
        s = _run(om, rx, iq, Kd, exhaustive=False, max_trials=cap)
        x = _run(om, rx, iq, Kd, exhaustive=True, max_trials=cap)
        assert np.array_equal(s[2], x[2]) and np.array_equal(s[1], x[1])
        for a, b in ((s[0], x[0]), (s[3], x[3])):
            assert np.array_equal(np.isfinite(a), np.isfinite(b))
            assert relerr(np.nan_to_num(a), np.nan_to_num(b)) < 2e-6
        tsr = s[2]
        assert tsr[5, 3] == 0 and not tsr[5].any() and not s[0][5].any()
        for f in (0, 1, 2, 3, 4, 6, 7):
            o = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, 100, 0.7, force_fp64=True)
            o.work(iq[f], np.zeros(fl, np.complex64))
            p_hit = int(o.time_synch_ref[0]) - cp
            if cap and p_hit >= cap:
                assert tsr[f, 3] == 0, (f, cap, p_hit)
            else:
                assert tsr[f, 3] == 1 and tsr[f, 0] == o.time_synch_ref[0] and tsr[f, 1] == o.time_synch_ref[1]


def test_screen_is_off_where_its_preconditions_fail(om):
    assert om.RxEngine(8, 64, 16, 62, (1, 3), 60, 100).set_sync_search(False) is True
    assert om.RxEngine(8, 64, 16, 60, (1, 3), 60, 100).set_sync_search(False) is False          # Ks != N - 2
    assert om.RxEngine(10, 64, 16, 62, (2, 3), 60, 100).set_sync_search(False) is False         # two sync symbols per pattern
    assert om.RxEngine(8, 64, 16, 62, (1, 3), 60, 100, compat=om.COMPAT_RXOFDM).set_sync_search(False) is False   # stride cp-1
    assert om.RxEngine(8, 64, 60, 62, (1, 3), 60, 100).set_sync_search(False) is False          # cp too long for one block


@pytest.mark.parametrize("N,cp,Kd,n_sym", [(64, 16, 60, 12), (256, 18, 152, 12), (2048, 144, 1200, 12),
                                            # long buffers: the search is staged (first 64 segments on a second stream under the
                                            # rest of the upload, the others behind them), short ones go through pinned memory
                                            (256, 18, 152, 400), (2048, 144, 1200, 60)])
def test_stream_block_uses_the_screened_search_with_the_same_outcome(om, N, cp, Kd, n_sym):
    """`work()` on host buffers (the GNU Radio path) searches, accepts and finalizes without a host decision in between.  A
    sequence of calls -- sync deep in the buffer, at sample 0, a buffer of noise only (nothing found: the previous estimate must
    stay), then a late sync again (in a long buffer: far behind the segments of the search's first stage) -- must leave the same
    report, output items and state rows as the exhaustive search and as the oracle."""
    L = N + cp
    leads = [L + 7, 0, None, (3 if n_sym < 100 else n_sym // 3) * L - 1, 17]
    rng = np.random.default_rng(N + 1)
    bufs = []
    for ld in leads:
        if ld is None:
            fl = n_sym * L
            bufs.append((0.2 * (rng.standard_normal(fl) + 1j * rng.standard_normal(fl))).astype(np.complex64))
        else:
            _, iq = _frames_with_leads(N, cp, Kd, n_sym, [ld], 0.02, seed=int(rng.integers(1 << 30)))
            bufs.append(iq[0][:n_sym * L].copy())
    scr = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7)
    exh = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7)
    assert scr.set_sync_search(False) is True and exh.set_sync_search(True) is False
    o = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, 100, 0.7, force_fp64=True)
    n_det = 0
    for i, x in enumerate(bufs):
        out_s, out_x, out_o = (np.zeros(len(x), np.complex64) for _ in range(3))
        n_s, n_x = scr.work(x, out_s), exh.work(x, out_x)
        n_o = o.work(x, out_o)
        rs, rx_ = scr.report, exh.report
        assert n_s == n_x == n_o, (i, n_s, n_x, n_o)
        for fld in ("detected", "trials_run", "count", "corr_obs", "n_data_items"):
            assert getattr(rs, fld) == getattr(rx_, fld), (i, fld)
        assert list(rs.time_synch_ref) == list(rx_.time_synch_ref), i
        assert rs.time_synch_ref[0] == o.time_synch_ref[0] and rs.time_synch_ref[1] == o.time_synch_ref[1], i
        assert rs.corr_obs == o.corr_obs, i
        n_det += rs.detected
        if leads[i] is None:
            assert not rs.detected
        if n_s > 0:
            ok = np.isfinite(out_o[:n_s]) & np.isfinite(out_s[:n_s])
            assert np.array_equal(np.isfinite(out_s[:n_s]), np.isfinite(out_x[:n_s]))
            assert relerr(out_s[:n_s][ok], out_x[:n_s][ok]) < 2e-6
            assert_close(out_s[:n_s][ok], out_o[:n_s][ok], "call %d" % i)
        for row in (0, 1):
            a, b = scr.state(row), exh.state(row)
            for k in ("chan_freq", "chan_time", "synch_freq", "eq_gain"):
                assert relerr(a[k], b[k]) < 2e-6, (i, row, k)
    assert n_det >= 3
