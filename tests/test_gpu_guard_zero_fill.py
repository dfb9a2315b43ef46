"""GPU parity of the rows the reference leaves at zero (run with -m gpu).

`SynchAndChanEst.py:88` creates `est_data_freq` as zeros and `:223` guards every [S, D] pattern ONCE: a pattern whose first data
window does not fit the buffer is skipped and its rows keep their zeros; the later data windows of a pattern that passed the guard
may run past the end and are zero-padded by `np.fft.fft(x, N)` (`:230`) -- or are empty, in which case the power normalisation
(`:233`) divides by zero and the row is NaN.  The batch path must WRITE all of that: every output buffer here is pre-filled with
0xFF, every row of every frame is compared (rows of zeros, zero-padded rows, NaN rows included) and the bits with the de-map of
the rows.  Frames: aligned; sync found later than 2L + cp (last pattern fails the guard); first sync symbol destroyed (the search
locks onto the second pattern); frame lengths cut to the guard's own boundary (passes by 0 samples / fails by 1)."""
import numpy as np
import pytest

from conftest import assert_close, poisoned
from oracle import ofdm_oracle as orc

pytestmark = pytest.mark.gpu

SIZES = [(64, 16, 60, 12), (256, 64, 180, 12), (512, 36, 300, 12), (1024, 72, 600, 12), (2048, 144, 1200, 12), (4096, 288, 2400, 8)]
MODES = [("QPSK", "unpacked"), ("QPSK", "packed"), ("16QAM", "packed"), ("64QAM", "packed"), ("64QAM", "unpacked")]


@pytest.fixture(scope="module")
def om():
    import ofdm_mi355x
    ofdm_mi355x.load()
    return ofdm_mi355x


def _frame(N, cp, Kd, n_sym, mod, lead, rng, fl, kill_first_sync=False, sigma=0.02):
    L = N + cp
    bps = orc.BITS_PER_SYMBOL[mod]
    bits = rng.integers(0, 2, (n_sym // 4) * 3 * Kd * bps).astype(np.uint8)
    tx = orc.channel_apply(orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym, modulation=mod), orc.REF_TAPS, N)
    if kill_first_sync:
        tx[:L] = 0.5 * (rng.standard_normal(L) + 1j * rng.standard_normal(L))
    pre = 0.3 * (rng.standard_normal(lead) + 1j * rng.standard_normal(lead))
    x = np.concatenate([pre, tx])[:fl]
    x = np.concatenate([x, np.zeros(fl - len(x))])
    return (x + sigma * (rng.standard_normal(fl) + 1j * rng.standard_normal(fl))).astype(np.complex64)


def _check_batch(om, N, cp, Kd, n_sym, mod, packed, iq, expect_zero_rows):
    """iq [n_frames, fl] through ofdm_rx_demod_frames into poisoned buffers; EVERY row against the oracle."""
    n_frames, fl = iq.shape
    bps = orc.BITS_PER_SYMBOL[mod]
    rx = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 30, 0.7, modulation=mod)
    rx.set_max_trials(0)
    nds = rx.data_symbols_per_frame(fl)
    nbits = n_frames * nds * Kd * bps
    d_iq = om.DeviceBuffer(iq.nbytes).upload(iq)
    d_eq = poisoned(om, n_frames * nds * Kd * 8)
    d_b = poisoned(om, nbits // 8 if packed else nbits)
    d_tsr = poisoned(om, n_frames * 16)
    assert rx.demod_frames(d_iq, n_frames, fl, fl, d_eq, d_b, om.BITS_PACKED if packed else om.BITS_UNPACKED, d_tsr) == nds
    eq = d_eq.download(np.complex64, n_frames * nds * Kd).reshape(n_frames, nds, Kd)
    b = d_b.download(np.uint8, nbits // 8 if packed else nbits)
    b = (np.unpackbits(b) if packed else b).reshape(n_frames, nds, Kd * bps)
    tsr = d_tsr.download(np.int32, n_frames * 4).reshape(n_frames, 4)
    assert b.max() <= 1, "bit rows left unwritten (0xFF poison)" if not packed else ""
    n_rows = (fl // (N + cp))
    keep = [r for r in range(max(n_sym, n_rows)) if r % 4 != 3][:nds]
    seen_zero = seen_pad = 0
    for f in range(n_frames):
        o = orc.RxOracle(max(n_sym, n_rows), N, cp, N - 2, [1, 3], Kd, 30, 0.7, force_fp64=True)
        with np.errstate(all="ignore"):
            o.work(iq[f], np.zeros(fl, np.complex64))
        assert tsr[f, 3] == 1 and tsr[f, 0] == o.time_synch_ref[0] and tsr[f, 1] == o.time_synch_ref[1], (f, tsr[f], o.time_synch_ref)
        ref = o.est_data_freq[keep]
        nan_ref = ~np.isfinite(ref).all(axis=1)
        nan_gpu = ~np.isfinite(eq[f]).all(axis=1)
        assert np.array_equal(nan_ref, nan_gpu), (f, nan_ref, nan_gpu)               # empty windows: 0/0 in both
        zero_ref = ~nan_ref & ~ref.any(axis=1)
        assert not eq[f][zero_ref].any(), "frame %d: rows %s must be zeros" % (f, np.nonzero(zero_ref)[0])
        live = ~nan_ref & ~zero_ref
        assert_close(eq[f][live], ref[live], "frame %d" % f)
        t0 = int(o.time_synch_ref[0])
        for r in np.nonzero(live)[0]:
            if t0 + (r // 3 * 4 + 1 + r % 3) * (N + cp) + N > fl:
                seen_pad += 1                                                        # a window np.fft.fft zero-padded
        seen_zero += int(zero_ref.sum())
        # bits: the de-map of the GPU's own rows (exact), and of the oracle's rows wherever the decision is not marginal
        ok = ~nan_ref
        assert np.array_equal(b[f][ok].ravel(), orc.demap_hard(eq[f][ok].ravel(), mod))
        zb = orc.demap_hard(np.zeros(Kd, np.complex64), mod)
        for r in np.nonzero(zero_ref)[0]:
            assert np.array_equal(b[f][r], zb)
    if expect_zero_rows:
        assert seen_zero >= 3, "the case did not make any pattern guard fail"
    return seen_zero, seen_pad


@pytest.mark.parametrize("mod,layout", MODES)
@pytest.mark.parametrize("N,cp,Kd,n_sym", SIZES)
def test_guard_failed_patterns_are_written_as_zeros(om, N, cp, Kd, n_sym, mod, layout):
    L = N + cp
    rng = np.random.default_rng(N * 7 + len(mod) + len(layout))
    fl = n_sym * L + 7
    iq = np.stack([
        _frame(N, cp, Kd, n_sym, mod, 0, rng, fl),                                   # aligned: every pattern fits
        _frame(N, cp, Kd, n_sym, mod, 2 * L + cp + 16, rng, fl),                     # last pattern fails the guard
        _frame(N, cp, Kd, n_sym, mod, 0, rng, fl, kill_first_sync=True),             # search locks onto the 2nd pattern (4L late)
        _frame(N, cp, Kd, n_sym, mod, 3 * L + 3, rng, fl),
        _frame(N, cp, Kd, n_sym, mod, L // 2, rng, fl),                              # guard passes, last window zero-padded (:230)
    ])
    zeros, padded = _check_batch(om, N, cp, Kd, n_sym, mod, layout == "packed", iq, expect_zero_rows=True)
    assert zeros >= 9 and padded >= 1                                                # three frames x one failed pattern


@pytest.mark.parametrize("mod,layout", [("QPSK", "unpacked"), ("16QAM", "packed")])
@pytest.mark.parametrize("N,cp,Kd,n_sym", SIZES)
def test_frame_length_at_the_guard_boundary(om, N, cp, Kd, n_sym, mod, layout):
    """`:223` reads `tsr0 + S*L*(P+1) + N - 1 <= len`: with the frame cut to exactly that length the last pattern is demodulated
    (its first window one sample short, the two after it empty or zero-padded); one sample shorter and it is skipped."""
    L = N + cp
    rng = np.random.default_rng(N * 11 + len(mod))
    long_fl = (n_sym + 4) * L
    x = _frame(N, cp, Kd, n_sym, mod, 2 * L + 2 * cp + 3, rng, long_fl)
    o = orc.RxOracle(n_sym + 4, N, cp, N - 2, [1, 3], Kd, 30, 0.7, force_fp64=True)
    with np.errstate(all="ignore"):
        o.work(x, np.zeros(long_fl, np.complex64))
    t0 = int(o.time_synch_ref[0])
    p_last = n_sym // 4 - 1
    fl_pass = t0 + L * (4 * p_last + 1) + N - 1
    assert fl_pass // L == n_sym                                                     # the last pattern is still counted (:140)
    z_pass, pad_pass = _check_batch(om, N, cp, Kd, n_sym, mod, layout == "packed", x[None, :fl_pass].copy(), expect_zero_rows=False)
    z_fail, _ = _check_batch(om, N, cp, Kd, n_sym, mod, layout == "packed", x[None, :fl_pass - 1].copy(), expect_zero_rows=True)
    assert z_pass == 0 and pad_pass >= 1 and z_fail == 3
