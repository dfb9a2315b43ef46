#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through the RX hot path (sync + LS estimate + CP strip + FFT + equalise + de-map)
on MI355X, with the kernel's HBM roofline fraction and a same-box CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfgA|n1024|n4096]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the receive chain over one batch of synthetic frames that is already resident in HBM
(bits -> HIP TX -> HIP channel, generated once, untimed).  N>1: every rank owns an equal shard of whole
frames (weak scaling, no data-path collective); the demodulated packed bit-stream is re-assembled on every
rank with one RCCL all-gather per sub-batch, overlapped with the demod of the next sub-batch.

Prints ONE JSON line on rank 0.  `value` counts every complex64 input sample consumed (CP and sync symbols
included), summed over all ranks, divided by the max-over-ranks time of K steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "lte-gnu-radio-code_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(nfft=2048, cp=144, Kd=1200, mod="16QAM", n_sym=240, frames=4369, chan="awgn", snr_db=30.0,
                 name="2048-pt FFT / 144-CP / Kd=1200 / 16-QAM, 4369 frames x 240 symbols (1,048,560 symbols), AWGN loopback"),
    # configs[2]
    "cfg3": dict(nfft=2048, cp=144, Kd=1200, mod="64QAM", n_sym=240, frames=4369, chan="rayleigh", snr_db=30.0, gate=0.3,
                 name="2048-pt FFT / 144-CP / Kd=1200 / 64-QAM, Rayleigh 8-tap per-frame fading + AWGN 30 dB"),
    # configs[0] shape (the reference's own CPU-runnable case), scaled up in frame count
    "cfgA": dict(nfft=64, cp=16, Kd=60, mod="QPSK", n_sym=240, frames=65536, chan="ref5tap", snr_db=100.0,
                 name="64-pt FFT / 16-CP / Kd=60 / QPSK, reference 5-tap channel"),
    "n1024": dict(nfft=1024, cp=72, Kd=600, mod="16QAM", n_sym=240, frames=8738, chan="awgn", snr_db=30.0,
                  name="1024-pt FFT / 72-CP / Kd=600 / 16-QAM, AWGN"),
    "n4096": dict(nfft=4096, cp=288, Kd=2400, mod="16QAM", n_sym=240, frames=2184, chan="awgn", snr_db=30.0,
                  name="4096-pt FFT / 288-CP / Kd=2400 / 16-QAM, AWGN"),
}
BPS = {"BPSK": 1, "QPSK": 2, "16QAM": 4, "64QAM": 6}


def build_inputs(torch, om, cfg, n_frames, device, seed):
    """bits -> HIP TX -> HIP channel, in sub-batches; returns (d_rx [n_frames*frame_len] complex64, packed TX bits)."""
    N, cp, Kd, mod, n_sym = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["mod"], cfg["n_sym"]
    L = N + cp
    fl = n_sym * L
    bps = BPS[mod]
    txe = om.TxEngine(N, cp, N - 2, Kd, (1, 3), mod, device=device)
    nbytes = txe.bits_per_frame(n_sym) // 8
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    bits = torch.randint(0, 256, (n_frames, nbytes), dtype=torch.uint8, device="cuda", generator=g)
    d_rx = torch.empty((n_frames, fl, 2), dtype=torch.float32, device="cuda")
    if cfg["chan"] == "awgn":
        taps = np.array([[1.0 + 0j]], np.complex64)
        per_frame = False
    elif cfg["chan"] == "ref5tap":
        t = np.array([0.3977, 0.7954 - 0.3977j, -0.1988, 0.0994, -0.0398])     # MultiAntennaSystem.py:64
        taps = (t / np.linalg.norm(t)).astype(np.complex64)[None, :]
        per_frame = False
    else:  # per-frame i.i.d. CN(0,p_l) taps, 8 taps, exponential profile, unit norm
        rng = np.random.default_rng(seed)
        p = np.exp(-np.arange(8) / 2.0)
        t = (rng.standard_normal((n_frames, 8)) + 1j * rng.standard_normal((n_frames, 8))) * np.sqrt(p / 2)
        taps = (t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.complex64)
        per_frame = True
    d_taps = torch.from_numpy(np.ascontiguousarray(taps).view(np.float32)).cuda()
    n_taps = taps.shape[1]
    # reference noise law (MultiAntennaSystem.py:244, 'Digital'): var = L/(Kd*bps) * sig_pow * 10^(-SNR/10), sig_pow = 1
    noise_var = (L / (Kd * bps)) * 10 ** (-cfg["snr_db"] / 10)
    stream = torch.cuda.current_stream().cuda_stream
    step = 256
    d_tx = torch.empty((min(step, n_frames), fl, 2), dtype=torch.float32, device="cuda")
    for f0 in range(0, n_frames, step):
        nf = min(step, n_frames - f0)
        txe.modulate_frames(bits[f0:f0 + nf], nf, n_sym, d_tx, fl, om.BITS_PACKED, stream)
        tp = d_taps[f0:f0 + nf] if per_frame else d_taps
        txe.channel(d_tx, nf, fl, fl, tp, n_taps, d_rx[f0:f0 + nf], fl, fl, noise_var=noise_var, seed=seed + f0,
                    per_frame_taps=per_frame, stream=stream)
    torch.cuda.synchronize()
    del d_tx
    return d_rx, bits


def cpu_baseline(cfg, iq_host):
    """oracle/cpu_baseline.py in a fresh child process (no GPU state) on a bounded sample of this workload."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sample.npy")
        np.save(path, iq_host)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), path,
                            json.dumps({k: cfg[k] for k in ("nfft", "cp", "Kd", "snr_db")})],
                           capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def _power_sample(torch, step, seconds=3.0):
    """[median package power W, median shader clock MHz] from rocm-smi while `step` runs back to back for `seconds` (untimed)."""
    import re
    import subprocess
    import threading
    samples, stop = [], [False]

    def sampler():
        time.sleep(1.0)
        while not stop[0]:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
            pw = re.search(r"Package Power \(W\): ([0-9.]+)", out)
            ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
            if pw and ck:
                samples.append((float(pw.group(1)), int(ck.group(1))))
            time.sleep(0.2)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    stop[0] = True
    th.join()
    if not samples:
        return None
    return [float(np.median([a for a, _ in samples])), float(np.median([b for _, b in samples]))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: the config's)")
    ap.add_argument("--no-eq", action="store_true", help="do not write equalised symbols (bits only)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--chunks", type=int, default=4, help="sub-batches per step for all-gather overlap (N>1)")
    ap.add_argument("--mod", default=None, choices=sorted(BPS), help="override the config's constellation (numerology sweep)")
    args = ap.parse_args()

    import torch
    import ofdm_mi355x as om
    from ofdm_mi355x import dist as od

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    # BENCH_REHEARSAL=1: rehearse the N>1 control flow on a box with fewer GPUs than ranks (ranks share devices, the
    # all-gather goes through gloo on host copies).  Never used for reported numbers.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    device = torch.cuda.current_device()
    # Everything runs on ONE explicit torch stream.  torch's default stream has handle 0, which the C ABI reads as "use the
    # handle's own (non-blocking) stream": kernels would then be unordered with torch's allocator, with torch copies and -- at
    # N > 1 -- with RCCL, which orders itself against torch's CURRENT stream only.
    torch.cuda.set_stream(torch.cuda.Stream(device=device))

    cfg = dict(CONFIGS[args.config])
    if args.mod:
        cfg["name"] = cfg["name"].replace(cfg["mod"].replace("QAM", "-QAM"), args.mod.replace("QAM", "-QAM"))
        cfg["mod"] = args.mod
    n_frames = args.frames or cfg["frames"]
    N, cp, Kd, mod, n_sym = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["mod"], cfg["n_sym"]
    L = N + cp
    fl = n_sym * L
    bps = BPS[mod]

    d_rx, tx_bits = build_inputs(torch, om, cfg, n_frames, device, seed=20260101 + rank)
    rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, cfg["snr_db"], cfg.get("gate", 0.7), modulation=mod, device=device)
    rxe.reserve(n_frames)
    rxe.set_max_trials(L)     # frames are generated aligned: never scan more than one symbol period for the sync
    rxe.set_profiling(True)
    nds = rxe.data_symbols_per_frame(fl)
    bytes_per_frame_bits = nds * Kd * bps // 8
    d_eq = None if args.no_eq else torch.empty((n_frames, nds, Kd, 2), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    bounds = od.sub_batches(n_frames, args.chunks if world > 1 else 1)
    n_chunks = len(bounds)
    # N>1: two generations of bit / receive buffers (ofdm_mi355x.dist.GatherPipeline): step i's all-gather runs under step i+1's
    # demod; the launch stream waits for a gather only when the buffers it read and wrote are about to be reused.
    pipe = od.GatherPipeline(dist, torch, world, bounds, n_frames, bytes_per_frame_bits, "cuda",
                             recv_device="cpu" if rehearsal else "cuda", host_staging=rehearsal)
    k_sync, k_demod = [], []

    def produce(bits, f0, f1):
        rxe.demod_frames(d_rx[f0:f1], f1 - f0, fl, fl, None if d_eq is None else d_eq[f0:f1], bits[f0:f1], om.BITS_PACKED, None, stream)

    def step():
        pipe.step(produce)

    drain = pipe.drain

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rxe.set_profiling(True)                   # resets the library's event ring: only the timed steps are averaged
    for _ in range(args.steps):
        step()
    drain()                                   # every all-gather of the timed steps completes inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    s_ms, d_ms = rxe.kernel_ms()              # HIP events recorded on the launch stream inside the timed region
    k_sync.append(s_ms)                       # (mean per launch; N>1 launches one sync+demod pair per sub-batch)
    k_demod.append(d_ms)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    samples_per_step = world * n_frames * fl
    ms_per_step = elapsed / args.steps * 1e3
    value = samples_per_step / (elapsed / args.steps) / 1e6

    # ---- correctness spot check of what was timed (rank 0): bit errors of frame 0 vs the transmitted bits
    ber = None
    d_bits, gathered = pipe.last()            # what the last step produced
    if rank == 0:
        rxb = d_bits[0].cpu().numpy()
        txb = tx_bits[0].cpu().numpy()
        ber = float(np.unpackbits(rxb ^ txb).sum()) / (len(rxb) * 8)
        if world > 1:
            assert torch.equal(od.reassemble(torch, gathered, world)[0].cpu(), d_bits.cpu()), "all-gather did not reassemble rank 0's own shard"

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (rx_demod_kernel): algorithmic bytes per launch / measured duration.
        # per data symbol the stream read is L*8 B (CP included), the write Kd*8 B (+ Kd*bps/8 B bits).
        roof = None
        if k_demod:
            dsym = n_frames * nds // n_chunks                      # data symbols per demod launch (rank 0)
            alg = dsym * (L * 8 + (0 if d_eq is None else Kd * 8) + Kd * bps // 8)
            dm = float(np.mean(k_demod))
            ach = alg / (dm * 1e-3) / 1e9
            traffic = None
            tf = os.path.join(ROOT, "profiles", "r01_demod_traffic.json")
            if world == 1 and os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get(args.config)
                except Exception:
                    traffic = None
            # same-run context: this chip's float4 device-copy rate and the demod access pattern without arithmetic
            from ofdm_mi355x import _lib as ol
            def _probe(mode):
                ts = []
                nb = d_eq.numel() * 4 if d_eq is not None else d_bits.numel()
                nb -= nb % 16
                dst = d_eq if d_eq is not None else d_bits
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ol.check(rxe.lib.ofdm_bandwidth_probe(device, ol.ptr(d_rx), ol.ptr(dst), nb, mode, N * 8, cp * 8, (Kd * 8) & ~15, dsym if d_eq is not None else 0, stream))
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1))
                return float(np.median(ts)), nb
            copy_gbs = pat_gbs = None
            if world == 1:
                cms, cnb = _probe(0)
                copy_gbs = 2 * cnb / cms / 1e6
                if d_eq is not None:
                    pms, _ = _probe(1)
                    pat_gbs = dsym * (L * 8 + Kd * 8) / pms / 1e6
            # same-run context: package power and shader clock while the step runs back to back (untimed extra loop; rocm-smi).
            # On the pool's chips this kernel sits at the ~1.3 kW package limit with the shader clock throttled below the
            # 2.4 GHz the memory-only probes run at.
            power = None
            profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
            if world == 1 and not profiled:     # no child processes under a profiler's preload
                try:
                    power = _power_sample(torch, step)
                except Exception:
                    power = None
            roof = dict(bound="hbm", kernel="rx_demod_kernel<%d>" % N, achieved=round(ach, 1), peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                        algorithmic_bytes_per_launch=int(alg), kernel_ms=round(dm, 4),
                        sync_kernel_ms=round(float(np.mean(k_sync)), 4),
                        read_only_frac=round(dsym * L * 8 / (dm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        measured_copy_GBs=None if copy_gbs is None else round(copy_gbs, 1),
                        access_pattern_no_math_GBs=None if pat_gbs is None else round(pat_gbs, 1),
                        package_power_w_and_sclk_mhz_under_load=power)
        cpu = None
        if world == 1 and not args.no_cpu:
            iq_host = d_rx[:16].cpu().numpy().view(np.complex64).reshape(16, fl)
            cpu = cpu_baseline(cfg, iq_host)
        out = {
            "metric": "IQ Msamples/s through RX FFT+equalize, 2048-pt/144-CP; %HBM roofline; 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "frames_per_gpu": n_frames, "symbols_per_gpu": n_frames * n_sym,
                       "samples_per_step": samples_per_step, "outputs": ("bits" if d_eq is None else "equalised symbols + packed bits"),
                       "parallelism": "frame-shard x%d%s" % (world, " + RCCL all-gather of packed bits (%d sub-batches per step, gather of step i under the demod of step i+1)" % n_chunks if world > 1 else ""),
                       "hbm_read_fraction_of_8TBs": round(value * 1e6 * 8 / world / 8e12, 4),
                       "bit_error_rate_frame0": ber},
            "roofline": roof, "cpu_baseline": cpu,
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
