#!/usr/bin/env python3
"""One-off differential campaign of the stream block (`work()` on host buffers, GPU): sequences of calls on ONE handle -- random
numerology, buffer length from a few symbols (pinned in-place path) to hundreds (staged segment search, split upload), sync at a
random depth, at sample 0, or nowhere (noise only) -- against the fp64 oracle's report, output items and state.
`python tools/fuzz_stream.py [seconds] [seed]`.  Not part of the test suite (its run time is the oracle's search)."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import ofdm_mi355x as om
from oracle import ofdm_oracle as orc
from conftest import assert_close, relerr

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
om.load()


def buffer_with_sync(N, cp, Kd, n_sym, lead, sigma, r):
    L = N + cp
    fl = n_sym * L
    if lead is None:
        return (0.2 * (r.standard_normal(fl) + 1j * r.standard_normal(fl))).astype(np.complex64)
    bits = r.integers(0, 2, (n_sym // 4) * 3 * Kd * 2).astype(np.uint8)
    tx = orc.channel_apply(orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym), orc.REF_TAPS, N)
    pre = 0.3 * (r.standard_normal(lead) + 1j * r.standard_normal(lead))
    x = np.concatenate([pre, tx])[:fl]
    x = np.concatenate([x, np.zeros(fl - len(x))])
    return (x + sigma * (r.standard_normal(fl) + 1j * r.standard_normal(fl))).astype(np.complex64)


def run_case(N, cp, Kd, n_sym, leads, sigma, seed):
    r = np.random.default_rng(seed)
    eng = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7)
    o = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, 100, 0.7, force_fp64=True)
    det = 0
    for i, ld in enumerate(leads):
        x = buffer_with_sync(N, cp, Kd, n_sym, ld, sigma, r)
        out_g, out_o = np.zeros(len(x), np.complex64), np.zeros(len(x), np.complex64)
        n_g = eng.work(x, out_g)
        with np.errstate(all="ignore"):
            n_o = o.work(x, out_o)
        rep = eng.report
        assert n_g == n_o, (i, n_g, n_o)
        assert rep.time_synch_ref[0] == o.time_synch_ref[0] and rep.time_synch_ref[1] == o.time_synch_ref[1], (i, list(rep.time_synch_ref), o.time_synch_ref)
        assert rep.corr_obs == o.corr_obs and rep.count == o.count, i
        det += rep.detected
        if n_g > 0:
            ok = np.isfinite(out_o[:n_g])
            assert np.array_equal(np.isfinite(out_g[:n_g]), ok), i
            if ok.any():
                assert_close(out_g[:n_g][ok], out_o[:n_g][ok], "call %d" % i)
        st = eng.state(0)
        assert relerr(st["chan_freq"], o.est_chan_freq_P[0]) < 1e-5, i
    return det


t0 = time.time()
n_cases = n_fail = n_det = 0
kinds = dict(small=0, mid=0, long=0)
while time.time() - t0 < budget:
    logn = int(rng.integers(6, 12))
    N = 1 << logn
    cp = {64: 16, 128: 9, 256: 18, 512: 36, 1024: 72, 2048: 144}[N] if rng.random() < 0.6 else int(rng.integers(max(4, N // 32), N // 8 + 1))
    Kd = min(int(rng.integers(max(2, N // 16), (N - 2) // 4 + 1)) * 4, (N - 2) // 4 * 4)
    L = N + cp
    kind = str(rng.choice(["small", "mid", "long"]))
    budget_trials = 4e8 / (N * logn)                               # oracle: one N-point FFT per trial before the sync
    if kind == "small":
        n_sym = int(rng.choice([4, 8]))
    elif kind == "mid":
        n_sym = int(rng.choice([12, 24, 32]))
    else:
        n_sym = 4 * max(3, int(min(600, budget_trials / L / 3)) // 4)
    kinds[kind] += 1
    calls = int(rng.integers(2, 5))
    leads = []
    for _ in range(calls):
        u = rng.random()
        if u < 0.15:
            leads.append(0)
        elif u < 0.25 and n_sym * L < budget_trials:
            leads.append(None)                                     # noise only: the oracle walks every trial
        else:
            leads.append(int(rng.integers(0, max(1, min((n_sym - 4) * L, int(budget_trials / calls))))))
    sigma = float(rng.choice([0.0, 0.02]))
    seed = int(rng.integers(1 << 31))
    n_cases += 1
    try:
        n_det += run_case(N, cp, Kd, n_sym, leads, sigma, seed)
    except Exception as e:
        n_fail += 1
        print("FAIL case", (N, cp, Kd, n_sym), "leads", leads, "sigma", sigma, "seed", seed, "->", type(e).__name__, str(e)[:300], flush=True)
        if n_fail <= 2:
            traceback.print_exc()
    if n_cases % 20 == 0:
        print("... %d cases, %d failed, %.0f s" % (n_cases, n_fail, time.time() - t0), flush=True)
print("fuzz_stream: %d cases %s, %d detections, %d failed" % (n_cases, kinds, n_det, n_fail))
sys.exit(1 if n_fail else 0)
