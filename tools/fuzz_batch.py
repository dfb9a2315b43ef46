#!/usr/bin/env python3
"""One-off differential campaign of the batch path (GPU): random numerology / bin count / prefix / frame count / leads / frame
length / constellation / bit layout, every row of every frame against the fp64 oracle through the poisoned-buffer checker of
tests/test_gpu_guard_zero_fill.py.  `python tools/fuzz_batch.py [seconds] [seed]`; prints one line per failing case and a summary.
Not part of the test suite (its run time is the oracle's)."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import ofdm_mi355x as om
from oracle import ofdm_oracle as orc
from conftest import assert_close, poisoned

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
om.load()


def frame(N, cp, Kd, n_sym, mod, lead, fl, kill, sigma, r):
    L = N + cp
    bps = orc.BITS_PER_SYMBOL[mod]
    bits = r.integers(0, 2, (n_sym // 4) * 3 * Kd * bps).astype(np.uint8)
    tx = orc.channel_apply(orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym, modulation=mod), orc.REF_TAPS, N)
    if kill:
        tx[:L] = 0.5 * (r.standard_normal(L) + 1j * r.standard_normal(L))
    pre = 0.3 * (r.standard_normal(lead) + 1j * r.standard_normal(lead))
    x = np.concatenate([pre, tx])[:fl]
    x = np.concatenate([x, np.zeros(fl - len(x))])
    return (x + sigma * (r.standard_normal(fl) + 1j * r.standard_normal(fl))).astype(np.complex64)


def run_case(c):
    N, cp, Kd, n_sym, mod, packed, n_frames, fl, leads, kills, sigma, cseed = c
    r = np.random.default_rng(cseed)
    bps = orc.BITS_PER_SYMBOL[mod]
    iq = np.stack([frame(N, cp, Kd, n_sym, mod, leads[f], fl, kills[f], sigma, r) for f in range(n_frames)])
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
    n_rows = fl // (N + cp)
    keep = [q for q in range(max(n_sym, n_rows)) if q % 4 != 3][:nds]
    zb = orc.demap_hard(np.zeros(Kd, np.complex64), mod)
    stats = dict(frames=n_frames, undetected=0, zero_rows=0, marginal=0)
    for f in range(n_frames):
        o = orc.RxOracle(max(n_sym, n_rows), N, cp, N - 2, [1, 3], Kd, 30, 0.7, force_fp64=True)
        with np.errstate(all="ignore"):
            o.work(iq[f], np.zeros(fl, np.complex64))
        if o.del_mat is not None and np.max(np.abs(o.del_mat)) > 0.7 * o.MM:        # the last trial evaluated is the accepted one
            m = np.max(np.abs(o.del_mat))
            if abs(m - 0.7 * o.MM) < 1e-3 * o.MM:
                stats["marginal"] += 1
                continue
            assert tsr[f, 3] == 1 and tsr[f, 0] == o.time_synch_ref[0] and tsr[f, 1] == o.time_synch_ref[1], (f, tsr[f], o.time_synch_ref)
        else:
            stats["undetected"] += 1
            assert tsr[f, 3] == 0, (f, tsr[f])
        ref = o.est_data_freq[keep]
        nan_ref = ~np.isfinite(ref).all(axis=1)
        nan_gpu = ~np.isfinite(eq[f]).all(axis=1)
        assert np.array_equal(nan_ref, nan_gpu), (f, nan_ref, nan_gpu)
        zero_ref = ~nan_ref & ~ref.any(axis=1)
        assert not eq[f][zero_ref].any(), "frame %d: rows %s must be zeros" % (f, np.nonzero(zero_ref)[0])
        live = ~nan_ref & ~zero_ref
        if live.any():
            assert_close(eq[f][live], ref[live], "frame %d" % f)
        ok = ~nan_ref
        assert np.array_equal(b[f][ok].ravel(), orc.demap_hard(eq[f][ok].ravel(), mod)), "bits of frame %d" % f
        for q in np.nonzero(zero_ref)[0]:
            assert np.array_equal(b[f][q], zb)
        stats["zero_rows"] += int(zero_ref.sum())
    return stats


t0 = time.time()
n_cases = n_fail = 0
tot = dict(frames=0, undetected=0, zero_rows=0, marginal=0)
sizes = {}
while time.time() - t0 < budget:
    logn = int(rng.integers(6, 13))
    N = 1 << logn
    cp = int(rng.integers(max(4, N // 32), N // 4 + 1))
    if rng.random() < 0.5:
        cp = {64: 16, 128: 9, 256: 18, 512: 36, 1024: 72, 2048: 144, 4096: 288}[N]
    Kd = int(rng.integers(max(2, N // 16), (N - 2) // 4 + 1)) * 4
    Kd = min(Kd, (N - 2) // 4 * 4)
    n_sym = int(rng.choice([8, 12, 16, 24]))
    mod = str(rng.choice(["QPSK", "16QAM", "64QAM"]))
    packed = bool(rng.random() < 0.6)
    L = N + cp
    # keep the oracle's share per case around a second or two: its search costs one N-point FFT per trial before the sync
    cap = max(1, int(5e8 // (1.5 * L * N * logn)))
    n_frames = int(rng.integers(1, min(24, cap) + 1))
    fl = n_sym * L + int(rng.integers(0, L))
    leads = [int(rng.integers(0, 3 * L)) if rng.random() < 0.7 else 0 for _ in range(n_frames)]
    kills = [bool(rng.random() < 0.1) for _ in range(n_frames)]
    sigma = float(rng.choice([0.0, 0.02, 0.1]))
    c = (N, cp, Kd, n_sym, mod, packed, n_frames, fl, leads, kills, sigma, int(rng.integers(1 << 31)))
    n_cases += 1
    sizes[N] = sizes.get(N, 0) + 1
    try:
        s = run_case(c)
        for k in tot:
            tot[k] += s[k]
    except Exception as e:
        n_fail += 1
        print("FAIL case", c[:8], "leads", c[8], "kills", c[9], "sigma", c[10], "seed", c[11], "->", type(e).__name__, str(e)[:300], flush=True)
        if n_fail <= 2:
            traceback.print_exc()
    if n_cases % 20 == 0:
        print("... %d cases, %d failed, %.0f s" % (n_cases, n_fail, time.time() - t0), flush=True)
print("fuzz_batch: %d cases (%s), %d failed; %s" % (n_cases, sizes, n_fail, tot))
sys.exit(1 if n_fail else 0)
