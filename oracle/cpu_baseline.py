#!/usr/bin/env python3
"""CPU baseline leg of bench.py (TEST/MEASUREMENT INFRASTRUCTURE, never on the product path).

Runs in its own process (no GPU state) on a bounded sample of the bench workload saved as .npy (whole frames of the batch
that was just timed on the GPU).  Four legs, BASELINE.md section 3 / SURVEY.md section 8(d):

 (i)   port        `rx_work_faithful_ops`: the reference block's OPERATION STRUCTURE (dense diag matmuls, per-symbol
                   np.fft.fft, Python loops) on whole 240-symbol frames, one frame after the other until the time budget is
                   spent (at most 4096 symbols); BLAS threads = whichever of {8, all} is faster.  This is `value`.
 (ii)  vectorised  `rx_demod_frames_vectorised`: honest batched-FFT NumPy, ONE WORKER PROCESS PER USABLE HOST CORE over disjoint
                   frames.  Workers are forked, load their frames and run one warm-up frame BEFORE the clock; every timed
                   repetition starts at a barrier and is the wall time from the first worker's start to the last worker's end;
                   >= 3 repetitions over >= 8 frames per worker, median reported.
 (iii) c_scalar    oracle/ofdm_oracle_c.c: the plain-C double-precision scalar restatement (sync search + LS estimate + demod of
                   a whole frame), single thread -- what one core does without NumPy's per-call overheads.
 (iv)  c_parallel  the same C on every usable core (OpenMP over frames, >= 8 frames per thread, 3 repetitions, median).

Prints one JSON object (flat keys, short strings).
"""
import json
import math
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ofdm_oracle as orc  # noqa: E402

FRAMES_PER_WORKER = 8
VEC_REPS = 3
MAX_SAMPLE_FRAMES = 512


def usable_cores():
    """(os.cpu_count(), cores this process may actually use): the scheduler affinity mask and a cgroup CPU quota both count."""
    host = os.cpu_count() or 1
    use = host
    try:
        use = min(use, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(p).read().split()
            if p.endswith("cpu.max"):
                if txt[0] != "max":
                    use = min(use, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    use = min(use, max(1, math.ceil(q / per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("BENCH_CPU_WORKERS")
    if env:
        use = max(1, int(env))
    return host, use


def sample_frames_wanted():
    return min(MAX_SAMPLE_FRAMES, max(16, usable_cores()[1] * FRAMES_PER_WORKER))


def _vec_worker(path, lo, hi, frame_len, ocfg, reps, barrier, q, wid):
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    iq = np.load(path, mmap_mode="r")
    mine = np.ascontiguousarray(iq[lo:hi]).reshape(-1)          # this worker's frames, in its own memory before the clock
    orc.rx_demod_frames_vectorised(mine[:frame_len], frame_len, ocfg)      # warm-up: imports, FFT plans, page faults
    for r in range(reps):
        barrier.wait()
        t0 = time.perf_counter()
        orc.rx_demod_frames_vectorised(mine, frame_len, ocfg)
        t1 = time.perf_counter()
        q.put((wid, r, t0, t1))


def main():
    path, cfg = sys.argv[1], json.loads(sys.argv[2])
    budget_s = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
    iq = np.load(path, mmap_mode="r")       # (n_frames, frame_len) complex64
    N, cp, Kd, snr = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["snr_db"]
    gate = cfg.get("gate", 0.7)
    L = N + cp
    n_avail, frame_len = iq.shape
    n_sym = frame_len // L
    host_cores, cores = usable_cores()

    # ---- (i) reference operation structure, whole frames
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        max_threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threadpool_limits = None
        max_threads = cores
    best = None
    settings = sorted({min(8, max_threads), max_threads})
    for nthr in settings:
        ctx = threadpool_limits(limits=nthr) if threadpool_limits else None
        t0 = time.perf_counter()
        frames_done = 0
        while True:
            frame = np.array(iq[frames_done % n_avail])
            rx = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, snr, gate)
            orc.rx_work_faithful_ops(rx, frame)
            assert rx.time_synch_ref[0] == cp, "the CPU sample is expected to be frame-aligned"
            frames_done += 1
            dt = time.perf_counter() - t0
            if dt > budget_s / len(settings) or frames_done * n_sym >= 4096:
                break
        if ctx is not None:
            ctx.restore_original_limits() if hasattr(ctx, "restore_original_limits") else ctx.unregister()
        rate = frames_done * frame_len / dt / 1e6
        if best is None or rate > best[0]:
            best = (rate, nthr, frames_done, dt)
    faithful, blas_threads, port_frames, port_dt = best

    # ---- (iii) plain C, one core, whole frames (fresh instance per frame: search + estimate + demod)
    c_rate = c_frames = None
    try:
        from oracle import oracle_c
        oracle_c.load()
        frame = np.array(iq[0])
        oracle_c.rx_work(frame, n_sym, N, cp, N - 2, (1, 3), Kd, snr, gate)          # warm-up
        t0 = time.perf_counter()
        c_frames = 0
        while True:
            frame = np.array(iq[c_frames % n_avail])
            tsr, _, _, _ = oracle_c.rx_work(frame, n_sym, N, cp, N - 2, (1, 3), Kd, snr, gate)
            assert tsr[0] == cp
            c_frames += 1
            c_dt = time.perf_counter() - t0
            if c_dt > budget_s / 4 or c_frames >= 64:
                break
        c_rate = c_frames * frame_len / c_dt / 1e6
    except Exception as e:          # a host without gcc: the other legs still stand
        c_rate, c_frames, c_dt = None, 0, 0.0
        sys.stderr.write("cpu_baseline: C leg skipped (%s: %s)\n" % (type(e).__name__, e))

    # ---- (iv) the same plain C on every usable core (OpenMP over frames, >= 8 frames per thread), three repetitions, median
    cpar_rate = cpar_min = cpar_max = None
    cpar_frames = 0
    try:
        from oracle import oracle_c
        cpar_frames = min(n_avail, cores * FRAMES_PER_WORKER)
        block = np.ascontiguousarray(iq[:cpar_frames])
        oracle_c.rx_work_frames(block[:cores], n_sym, N, cp, N - 2, (1, 3), Kd, snr, gate, n_threads=cores)      # warm-up
        rates = []
        for _ in range(VEC_REPS):
            t0 = time.perf_counter()
            tsr_all, hits = oracle_c.rx_work_frames(block, n_sym, N, cp, N - 2, (1, 3), Kd, snr, gate, n_threads=cores)
            rates.append(cpar_frames * frame_len / (time.perf_counter() - t0) / 1e6)
            assert hits == cpar_frames and (tsr_all[:, 0] == cp).all()
        rates.sort()
        cpar_rate, cpar_min, cpar_max = rates[len(rates) // 2], rates[0], rates[-1]
    except Exception as e:
        sys.stderr.write("cpu_baseline: parallel C leg skipped (%s: %s)\n" % (type(e).__name__, e))

    # ---- (ii) vectorised NumPy, one worker per usable core, warmed before the clock
    ocfg = dict(nfft=N, cp_len=cp, synch_dat=(1, 3), num_synch_bins=N - 2, num_data_bins=Kd, snr=snr, scale_factor_gate=gate)
    per = FRAMES_PER_WORKER
    workers = cores
    disjoint = workers * per <= n_avail
    ctx = mp.get_context("fork")
    barrier = ctx.Barrier(workers + 1)
    q = ctx.Queue()
    procs = []
    for w in range(workers):
        lo = (w * per) % max(n_avail - per + 1, 1) if not disjoint else w * per
        p = ctx.Process(target=_vec_worker, args=(path, lo, lo + per, frame_len, ocfg, VEC_REPS, barrier, q, w))
        p.start()
        procs.append(p)
    rep_wall = []
    for r in range(VEC_REPS):
        barrier.wait(timeout=600)               # every worker is loaded and warm: the repetition starts now
        got = [q.get(timeout=600) for _ in range(workers)]
        rep_wall.append(max(g[3] for g in got) - min(g[2] for g in got))
    for p in procs:
        p.join(timeout=60)
    vec_samples = workers * per * frame_len
    vec_rates = sorted(vec_samples / w / 1e6 for w in rep_wall)
    vec_median = vec_rates[len(vec_rates) // 2]

    print(json.dumps(dict(
        value=round(faithful, 4), unit="Msamples/s", cores=int(blas_threads), kind="port", host_cores=int(host_cores),
        usable_cores=int(cores),
        sample="%d frames x %d symbols (%d symbols), reference op structure, %.1f s" % (port_frames, n_sym, port_frames * n_sym, port_dt),
        vectorised_value=round(vec_median, 2), vectorised_min=round(vec_rates[0], 2), vectorised_max=round(vec_rates[-1], 2),
        vectorised_cores=int(workers), vectorised_reps=int(VEC_REPS), vectorised_frames_per_worker=int(per),
        vectorised_sample="%d workers x %d %sframes x %d reps, warmed, median" % (workers, per, "disjoint " if disjoint else "", VEC_REPS),
        c_scalar_value=None if c_rate is None else round(c_rate, 2), c_scalar_cores=1,
        c_scalar_sample="%d frames x %d symbols, plain C fp64, 1 thread, %.1f s" % (c_frames, n_sym, c_dt),
        c_parallel_value=None if cpar_rate is None else round(cpar_rate, 2),
        c_parallel_min=None if cpar_min is None else round(cpar_min, 2), c_parallel_max=None if cpar_max is None else round(cpar_max, 2),
        c_parallel_cores=int(cores),
        c_parallel_sample="%d frames x %d symbols, plain C fp64, %d OpenMP threads, %d reps, median" % (cpar_frames, n_sym, cores, VEC_REPS))))


if __name__ == "__main__":
    main()
