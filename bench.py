#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through the RX hot path (sync + LS estimate + CP strip + FFT + equalise + de-map)
on MI355X, with the kernel's HBM roofline fraction and a same-box CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfgA|n128|n256|n512|n1024|n4096] [--lead random]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher: before anything touches the
GPU it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child and
relays rank 0's JSON line and the exit code.  Started by torchrun itself (WORLD_SIZE set) it is one rank.

A "step" = one pass of the receive chain over the batch(es) of synthetic frames that are already resident in HBM
(bits -> HIP TX -> HIP channel, generated once, untimed).  N>1: every rank owns an equal shard of whole frames (weak
scaling, no data-path collective); the demodulated packed bit-stream is re-assembled on every rank with one RCCL
all-gather per sub-batch, overlapped with the demod of the next sub-batch.

Prints ONE JSON line on rank 0.  `value` counts every complex64 input sample consumed (CP and sync symbols included),
summed over all ranks, divided by the max-over-ranks time of K steps.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "lte-gnu-radio-code_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
METRIC = "IQ Msamples/s through RX FFT+equalize, 2048-pt/144-CP; %HBM roofline; 1/2/4/8 GPU"

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(nfft=2048, cp=144, Kd=1200, mod="16QAM", n_sym=240, frames=4369, chan="awgn", snr_db=30.0,
                 name="2048-pt FFT / 144-CP / Kd=1200 / 16-QAM, 4369 frames x 240 symbols (1,048,560 symbols), AWGN loopback"),
    # configs[2]
    "cfg3": dict(nfft=2048, cp=144, Kd=1200, mod="64QAM", n_sym=240, frames=4369, chan="rayleigh", snr_db=30.0, gate=0.3,
                 name="2048-pt FFT / 144-CP / Kd=1200 / 64-QAM, Rayleigh 8-tap per-frame fading + AWGN 30 dB"),
    # configs[3]: 64 Mi symbols on 8 GPUs = 8 batches of 1 Mi symbols per GPU, all resident; one step = all 8 batches,
    # the bits of every batch all-gathered while the next batch is demodulated
    "cfg4": dict(nfft=2048, cp=144, Kd=1200, mod="64QAM", n_sym=240, frames=4369, batches=8, chan="awgn", snr_db=30.0,
                 name="2048-pt FFT / 144-CP / Kd=1200 / 64-QAM, 8 batches x 4369 frames x 240 symbols per GPU "
                      "(8,388,480 symbols per GPU; 64 Mi symbols on 8 GPUs), AWGN, all-gather of the packed bits per batch"),
    # configs[0] shape (the reference's own CPU-runnable case), scaled up in frame count
    "cfgA": dict(nfft=64, cp=16, Kd=60, mod="QPSK", n_sym=240, frames=65536, chan="ref5tap", snr_db=100.0,
                 name="64-pt FFT / 16-CP / Kd=60 / QPSK, reference 5-tap channel"),
    "n128": dict(nfft=128, cp=9, Kd=72, mod="16QAM", n_sym=240, frames=32768, chan="awgn", snr_db=30.0,
                 name="128-pt FFT / 9-CP / Kd=72 / 16-QAM, AWGN"),
    "n256": dict(nfft=256, cp=18, Kd=180, mod="16QAM", n_sym=240, frames=16384, chan="awgn", snr_db=30.0,
                 name="256-pt FFT / 18-CP / Kd=180 / 16-QAM, AWGN"),
    "n512": dict(nfft=512, cp=36, Kd=300, mod="16QAM", n_sym=240, frames=16384, chan="awgn", snr_db=30.0,
                 name="512-pt FFT / 36-CP / Kd=300 / 16-QAM, AWGN"),
    "n1024": dict(nfft=1024, cp=72, Kd=600, mod="16QAM", n_sym=240, frames=8738, chan="awgn", snr_db=30.0,
                  name="1024-pt FFT / 72-CP / Kd=600 / 16-QAM, AWGN"),
    "n4096": dict(nfft=4096, cp=288, Kd=2400, mod="16QAM", n_sym=240, frames=2184, chan="awgn", snr_db=30.0,
                  name="4096-pt FFT / 288-CP / Kd=2400 / 16-QAM, AWGN"),
}
BPS = {"BPSK": 1, "QPSK": 2, "16QAM": 4, "64QAM": 6}

KERNEL_SOURCES = ("rx_demod.hpp", "fft_core.hpp", "ofdm_device.hpp", "ofdm_launch.hpp")


def kernel_source_sha():
    """Identity of the demod kernel's source: a committed PMC traffic figure is only quoted for the build it was measured on."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(PKG, "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_inputs(torch, om, cfg, n_frames, device, seed, lead="aligned"):
    """bits -> HIP TX -> HIP channel, in sub-batches.  Returns (d_rx [n_frames, frame_len, 2] float32, packed TX bits, leads).
    lead='random': every frame is preceded by its own random number (0..L-1) of noise-only samples (same frame length: the
    tail of the frame is cut), so the sync search of every frame ends at a different trial."""
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
        taps = rayleigh_taps(n_frames, seed)
        per_frame = True
    d_taps = torch.from_numpy(np.ascontiguousarray(taps).view(np.float32)).cuda()
    n_taps = taps.shape[1]
    # reference noise law (MultiAntennaSystem.py:244, 'Digital'): var = L/(Kd*bps) * sig_pow * 10^(-SNR/10), sig_pow = 1
    noise_var = (L / (Kd * bps)) * 10 ** (-cfg["snr_db"] / 10)
    leads = None
    if lead != "aligned":
        leads = np.random.default_rng(seed + 7).integers(0, L, n_frames) if lead == "random" else np.full(n_frames, int(lead))
        d_rx.normal_(0.0, float(np.sqrt(noise_var / 2)), generator=g)         # the lead samples are noise only
    stream = torch.cuda.current_stream().cuda_stream
    step = 256
    d_tx = torch.empty((min(step, n_frames), fl, 2), dtype=torch.float32, device="cuda")
    for f0 in range(0, n_frames, step):
        nf = min(step, n_frames - f0)
        txe.modulate_frames(bits[f0:f0 + nf], nf, n_sym, d_tx, fl, om.BITS_PACKED, stream)
        if leads is None:
            tp = d_taps[f0:f0 + nf] if per_frame else d_taps
            txe.channel(d_tx, nf, fl, fl, tp, n_taps, d_rx[f0:f0 + nf], fl, fl, noise_var=noise_var, seed=seed + f0,
                        per_frame_taps=per_frame, stream=stream)
        else:                                 # one channel launch per frame, written `lead` samples into the frame
            for i in range(nf):
                ld = int(leads[f0 + i])
                tp = d_taps[f0 + i] if per_frame else d_taps
                txe.channel(d_tx[i], 1, fl, fl, tp, n_taps, d_rx[f0 + i, ld:], fl - ld, fl - ld, noise_var=noise_var,
                            seed=seed + f0 + i, per_frame_taps=False, stream=stream)
    torch.cuda.synchronize()
    del d_tx
    return d_rx, bits, leads


def rayleigh_taps(n_frames, seed):
    rng = np.random.default_rng(seed)
    p = np.exp(-np.arange(8) / 2.0)
    t = (rng.standard_normal((n_frames, 8)) + 1j * rng.standard_normal((n_frames, 8))) * np.sqrt(p / 2)
    return (t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.complex64)


def _clean_env():
    """Environment for child processes that must not inherit a profiler's preload."""
    env = dict(os.environ)
    for k in list(env):
        if k == "LD_PRELOAD" or k.startswith("ROCPROF") or k.startswith("ROCP_") or k.startswith("ROCTX"):
            env.pop(k)
    return env


def _profiled():
    return "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)


def cpu_baseline(cfg, d_rx, fl):
    """oracle/cpu_baseline.py in a fresh child process (no GPU state) on a bounded sample of this workload: whole frames of the
    batch that was just timed, >= 8 per usable host core for the vectorised leg (oracle/cpu_baseline.py names its four legs)."""
    import tempfile
    from oracle import cpu_baseline as cb
    n = min(int(d_rx.shape[0]), cb.sample_frames_wanted())
    iq_host = d_rx[:n].cpu().numpy().view(np.complex64).reshape(n, fl)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sample.npy")
        np.save(path, iq_host)
        del iq_host
        keys = {k: cfg[k] for k in ("nfft", "cp", "Kd", "snr_db")}
        keys["gate"] = cfg.get("gate", 0.7)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), path, json.dumps(keys)],
                           capture_output=True, text=True, timeout=900, env=_clean_env())
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def _power_sample(torch, step, seconds=3.0):
    """[median package power W, median shader clock MHz] from rocm-smi while `step` runs back to back for `seconds` (untimed)."""
    import re
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


def _rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(map(str, v)) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------ launcher (N > 1)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Parent of an N-rank run.  Touches no GPU (torch.cuda.device_count() only counts devices on this image): the ranks are
    fresh processes started by torch.distributed.run, one per GPU; their rank 0 prints the JSON line, which is relayed."""
    if not (args.dry_launch or os.environ.get("BENCH_REHEARSAL") == "1"):
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: this node exposes %d GPU(s). One rank per GPU is required "
                             "(BENCH_REHEARSAL=1 rehearses the control flow with ranks sharing devices; never a measurement)."
                             % (args.gpus, have))
    env = _clean_env() if _profiled() else dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(rc)
    if line is None:
        raise SystemExit("bench.py: the ranks exited without printing a result line")
    got = json.loads(line)["n_gpus"]
    if got != args.gpus:
        raise SystemExit("bench.py: asked for %d ranks, the result line reports %d" % (args.gpus, got))
    print(line)
    return 0


def dry_launch_rank(args, world, rank):
    """--dry-launch: the N-rank control flow (rendezvous, shard, GatherPipeline, max-over-ranks timing, result line) on CPU
    tensors over gloo with a fake `produce`.  No GPU, no kernel, no claim: the line says so.  Used by tests/test_bench_launch.py."""
    import torch
    import torch.distributed as dist
    from ofdm_mi355x import dist as od
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
    cfg = CONFIGS[args.config]
    n_frames, row_bytes = 64, 96
    batches = cfg.get("batches", 1)
    bounds = od.sub_batches(n_frames * batches, max(args.chunks if world > 1 else 1, batches))
    pipe = od.GatherPipeline(dist if world > 1 else None, torch, world, bounds, n_frames * batches, row_bytes, "cpu",
                             generations=1 if batches > 1 else 2, algo="direct" if args.gather == "direct" else "collective", rank=rank)

    def produce(bits, f0, f1):
        bits[f0:f1] = (torch.arange(f0, f1, dtype=torch.int64)[:, None] * 7 + rank * 31 + torch.arange(row_bytes)[None, :]).to(torch.uint8)

    for _ in range(args.warmup):
        pipe.step(produce)
    pipe.drain()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step(produce)
    pipe.drain()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        bits, gathered = pipe.last()
        whole = od.reassemble(torch, gathered, world)
        assert torch.equal(whole[rank], bits), "all-gather did not reassemble this rank's own shard"
        other = (rank + 1) % world
        exp = torch.empty_like(bits)
        f = torch.arange(0, n_frames * batches, dtype=torch.int64)[:, None]
        exp[:] = (f * 7 + other * 31 + torch.arange(row_bytes)[None, :]).to(torch.uint8)
        assert torch.equal(whole[other], exp), "rank %d holds wrong bits of rank %d" % (rank, other)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "dry-launch: control flow only, no GPU work, not a measurement",
                          "config": {"workload": cfg["name"], "world_size": world, "sub_batches_per_step": len(bounds)},
                          "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------ one rank
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed steps (default 250: >= 1 s of timed region at cfg2)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU and batch (default: the config's)")
    ap.add_argument("--no-eq", action="store_true", help="do not write equalised symbols (bits only)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-probes", action="store_true", help="skip the copy / access-pattern / power context probes")
    ap.add_argument("--chunks", type=int, default=4, help="sub-batches per step for all-gather overlap (N>1)")
    ap.add_argument("--mod", default=None, choices=sorted(BPS), help="override the config's constellation (numerology sweep)")
    ap.add_argument("--lead", default="aligned",
                    help="aligned | random (every frame starts after its own 0..L-1 sample lead: the sync search of every frame ends "
                         "at a different trial) | <int> (the same lead for every frame; kernel studies)")
    ap.add_argument("--sync-search", default="screened", choices=["screened", "exhaustive"],
                    help="exhaustive: the reference's trial-by-trial sync search (A/B against the screened search; same outputs)")
    ap.add_argument("--gather", default="auto", choices=["auto", "collective", "direct"],
                    help="N > 1: re-assembly of the packed bits as one collective (all_gather_into_tensor), as a group of point-to-point "
                         "transfers (one per xGMI link and direction), or whichever of the two an untimed calibration before the timed "
                         "loop finds faster on this node (default)")
    ap.add_argument("--dry-launch", action="store_true", help="rehearse the N-rank control flow on CPU/gloo (tests); no GPU work")
    ap.add_argument("--contention-probe", action="store_true",
                    help="N = 1 only, after the timed loop: the same steps again while a second stream moves what an 8-GPU all-gather "
                         "would put on this GPU's HBM (7 x the step's packed bits written, the same amount read); untimed for `value`")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        return launch_ranks(args, argv)
    world = int(world_env) if world_env is not None else 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_launch:
        return dry_launch_rank(args, world, rank)

    import torch
    import ofdm_mi355x as om
    from ofdm_mi355x import dist as od

    dist = None
    # BENCH_REHEARSAL=1: rehearse the N>1 control flow on a box with fewer GPUs than ranks (ranks share devices, the
    # all-gather goes through gloo on host copies).  Never used for reported numbers.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    # BENCH_FORCE_DIST=1 with one rank: a one-rank RCCL group runs every collective of the N > 1 path (init, asynchronous
    # all_gather_into_tensor on the launch stream, waits, all_reduce, barrier) -- the only way to execute those calls on a box
    # with one GPU (RCCL refuses two ranks on one device).  The line it prints is not a measurement of anything.
    force_dist = world == 1 and os.environ.get("BENCH_FORCE_DIST") == "1"
    multi = world > 1 or force_dist
    if force_dist:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() < world:
                raise SystemExit("WORLD_SIZE=%d but %d GPU(s) visible" % (world, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout carries the one JSON line, so
            # file descriptor 1 points at stderr until the communicator exists
            sys.stdout.flush()
            fd1 = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(fd1, 1)
                os.close(fd1)
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
    if args.lead == "random":
        cfg["name"] += ", per-frame random lead 0..L-1"
    elif args.lead != "aligned":
        cfg["name"] += ", lead %d" % int(args.lead)
    n_frames = args.frames or cfg["frames"]
    batches = cfg.get("batches", 1)
    N, cp, Kd, mod, n_sym = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["mod"], cfg["n_sym"]
    L = N + cp
    fl = n_sym * L
    bps = BPS[mod]

    # Pre-flight: the whole plan is resident at once (cfg4 at N = 8: 147 GB IQ + 7.5 GB equalised symbols + 5.7 GB bits + 45 GB
    # receive buffer = 205 GB of the 288).  Check it against what the device reports free BEFORE the first allocation, loudly.
    nds_plan = (n_sym // 4) * 3
    bits_rows = n_frames * batches * (nds_plan * Kd * bps // 8)
    gens = 1 if batches > 1 else 2
    plan = dict(iq=batches * n_frames * fl * 8, tx_staging=min(256, n_frames) * fl * 8, tx_bits=bits_rows,
                eq=0 if args.no_eq else n_frames * nds_plan * Kd * 8, bits=gens * bits_rows,
                gather_recv=(gens * world * bits_rows) if (world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1") else 0,
                frame_state=n_frames * (2 * N + Kd) * 8 + n_frames * 16)
    need = int(sum(plan.values()) * 1.02) + (1 << 30)          # 2 % allocator rounding + 1 GiB for the context, RCCL and probes
    free_b, total_b = torch.cuda.mem_get_info()
    if need > free_b:
        raise SystemExit("bench.py --config %s --gpus %d: the resident plan needs %.1f GB per GPU (%s) but device %d reports %.1f GB free "
                         "of %.1f GB.  Nothing was allocated." % (args.config, world, need / 1e9,
                         ", ".join("%s %.1f" % (k, v / 1e9) for k, v in plan.items() if v), device, free_b / 1e9, total_b / 1e9))
    inputs = [build_inputs(torch, om, cfg, n_frames, device, seed=20260101 + rank + 1000 * b, lead=args.lead) for b in range(batches)]
    rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, cfg["snr_db"], cfg.get("gate", 0.7), modulation=mod, device=device)
    rxe.reserve(n_frames)
    # aligned frames: the sync sits at trial 0 by construction, never scan more than one symbol period; random leads: the
    # lead is < L, two symbol periods cover it
    rxe.set_max_trials(L if args.lead == "aligned" else 2 * L)
    sync_mode = "screened" if rxe.set_sync_search(args.sync_search == "exhaustive") else "exhaustive"
    rxe.set_profiling(True)
    nds = rxe.data_symbols_per_frame(fl)
    bytes_per_frame_bits = nds * Kd * bps // 8
    d_eq = None if args.no_eq else torch.empty((n_frames, nds, Kd, 2), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    per_batch = args.chunks if (multi and batches == 1) else 1
    bounds = []
    for b in range(batches):                     # rows of the step's bit buffer: batch-major, frames within
        bounds += [(b * n_frames + f0, b * n_frames + f1) for f0, f1 in od.sub_batches(n_frames, per_batch)]
    n_chunks = len(bounds)
    # N>1: step i's all-gathers run under later demod launches (ofdm_mi355x.dist.GatherPipeline).  One batch per step: two
    # generations of bit / receive buffers, the gather of step i hides under step i+1.  Several batches per step (cfg4): one
    # generation is enough -- the gather of batch b has the other batches' demod time before its buffers come round again.
    pipe = od.GatherPipeline(dist, torch, world, bounds, n_frames * batches, bytes_per_frame_bits, "cuda",
                             recv_device="cpu" if rehearsal else "cuda", host_staging=rehearsal,
                             generations=1 if batches > 1 else 2, gather_at_world_1=force_dist,
                             algo="direct" if args.gather == "direct" else "collective", rank=rank)
    # --gather auto: which spelling of the re-assembly is faster is a property of the node (RCCL's all-gather algorithm against
    # 2(W-1) point-to-point transfers, one per link and direction).  Both are run on the first sub-batch's buffers before anything
    # is timed, every rank contributes its slower time, and the faster form is used for the timed loop.  A failure of the
    # point-to-point form on this node simply leaves the collective in place.
    gather_calibration = None
    if multi and pipe.gather and args.gather == "auto":
        try:
            f0, f1 = bounds[0]
            src = pipe.bits[0][f0:f1].cpu() if rehearsal else pipe.bits[0][f0:f1].contiguous()
            src.zero_()
            times = {}
            for algo in ("collective", "direct"):
                pipe.algo = algo
                for it in range(5):
                    if it == 2:
                        torch.cuda.synchronize()
                        dist.barrier()
                        t_c = time.perf_counter()
                    pipe.gather_rows(pipe.recv[0][0], src).wait()
                torch.cuda.synchronize()
                tt = torch.tensor([(time.perf_counter() - t_c) / 3 * 1e3], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                times[algo] = float(tt.item())
            pipe.algo = "direct" if times["direct"] < 0.97 * times["collective"] else "collective"
            gather_calibration = dict(collective_ms=round(times["collective"], 4), direct_ms=round(times["direct"], 4), chosen=pipe.algo,
                                      bytes_per_rank=int(src.numel()))
        except Exception as e:
            pipe.algo = "collective"
            gather_calibration = dict(error="%s: %s" % (type(e).__name__, e), chosen="collective")

    def produce(bits, r0, r1):
        b, f0 = divmod(r0, n_frames)
        f1 = f0 + (r1 - r0)
        rxe.demod_frames(inputs[b][0][f0:f1], f1 - f0, fl, fl, None if d_eq is None else d_eq[f0:f1], bits[r0:r1], om.BITS_PACKED, None, stream)

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
    # one event per step boundary on the launch stream (torch's current stream IS the stream the C ABI launches on): the
    # distribution of the step time (median, min: SURVEY 8(d)) beside the wall-clock mean that `value` is computed from
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    drain()                                   # every all-gather of the timed steps completes inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_sync, k_demod = rxe.kernel_ms()         # HIP events recorded on the launch stream inside the timed region
    if dist is not None:                      # (mean per launch over the last <= 32 launches)
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    samples_per_step = world * batches * n_frames * fl
    ms_per_step = elapsed / args.steps * 1e3
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])      # rank 0's launch stream
    value = samples_per_step / (elapsed / args.steps) / 1e6

    # ---- correctness spot check of what was timed (rank 0): bit errors of frame 0 vs the transmitted bits
    ber = None
    gather_info = None
    d_bits, gathered = pipe.last()            # what the last step produced
    if rank == 0:
        lead0 = 0 if inputs[0][2] is None else int(inputs[0][2][0])
        rxb = d_bits[0].cpu().numpy()
        txb = inputs[0][1][0].cpu().numpy()
        if lead0 == 0:
            ber = float(np.unpackbits(rxb ^ txb).sum()) / (len(rxb) * 8)
        else:                                 # a lead pushes the last pattern out of the frame: compare the patterns that fit
            keep = (len(rxb) // (n_sym // 4)) * ((fl - lead0) // (4 * L))
            ber = float(np.unpackbits(rxb[:keep] ^ txb[:keep]).sum()) / max(keep * 8, 1)
    if multi:
        assert torch.equal(od.reassemble(torch, gathered, world)[rank].cpu(), d_bits.cpu()), "all-gather did not reassemble this rank's own shard"
        # one all-gather of a sub-batch on its own (untimed extra): what the fabric gives without demod traffic beside it
        f0, f1 = bounds[0]
        src = d_bits[f0:f1].cpu() if rehearsal else d_bits[f0:f1].contiguous()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(5):
            pipe.gather_rows(gathered[0], src).wait()
        torch.cuda.synchronize()
        g_ms = (time.perf_counter() - tg) / 5 * 1e3
        nb = src.numel()
        # SURVEY 8(e) "report both": the same step with bit-error COUNTS exchanged instead of the bits -- every rank counts its
        # shard's errors against the transmitted bits on the device and the ranks all-reduce 8 bytes.  Not `value`: it shows what
        # the frame shards give when the fabric carries nothing.
        counts_only = None
        try:
            k2 = max(1, min(args.steps, 50))
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            tx_bits = [inp[1] for inp in inputs]

            def count_step():
                bits = pipe.bits[0]
                cnt.zero_()
                for (r0, r1) in bounds:
                    produce(bits, r0, r1)
                    b, f0 = divmod(r0, n_frames)
                    om.count_bit_errors(bits[r0:r1], tx_bits[b][f0:f0 + (r1 - r0)], (r1 - r0) * bytes_per_frame_bits, cnt, stream)
                tot = cnt.cpu() if rehearsal else cnt.clone()
                dist.all_reduce(tot)
                return tot
            count_step()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for _ in range(k2):
                tot = count_step()
            torch.cuda.synchronize()
            dist.barrier()
            c_ms = (time.perf_counter() - tc) / k2 * 1e3
            tcm = torch.tensor([c_ms], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tcm, op=dist.ReduceOp.MAX)
            c_ms = float(tcm.item())
            counts_only = dict(what="same step, bit-error counts all-reduced (8 B per rank) instead of the bits all-gathered",
                               steps=k2, ms_per_step=round(c_ms, 4), value_Msamples_per_s=round(samples_per_step / (c_ms * 1e-3) / 1e6, 1),
                               bit_errors_all_ranks=int(tot.item()), bits_all_ranks=int(d_bits.numel()) * 8 * world)
        except Exception as e:                      # never lose the measured line to the extra leg
            counts_only = dict(error="%s: %s" % (type(e).__name__, e))
        gather_info = dict(world_size=dist.get_world_size(), backend=dist.get_backend(), algorithm=pipe.algo, calibration=gather_calibration,
                           rccl_version=_rccl_version(torch) if not rehearsal else None,
                           bytes_contributed_per_rank_and_step=int(d_bits.numel()), bytes_received_per_rank_and_step=int(d_bits.numel() * (world - 1)),
                           allgather_alone_ms=round(g_ms, 4), allgather_alone_busbw_GBs=round(nb * (world - 1) / (g_ms * 1e-3) / 1e9, 2),
                           allgather_in_loop_GBs_received_per_rank=round(d_bits.numel() * (world - 1) / (ms_per_step * 1e-3) / 1e9, 2),
                           counts_only=counts_only)

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (rx_demod_kernel): algorithmic bytes per launch / measured duration.
        # per data symbol the stream read is L*8 B (CP included), the write Kd*8 B (+ Kd*bps/8 B bits).
        dsym = (bounds[0][1] - bounds[0][0]) * nds                  # data symbols per demod launch (rank 0)
        alg = dsym * (L * 8 + (0 if d_eq is None else Kd * 8) + Kd * bps // 8)
        dm = float(k_demod)
        ach = alg / (dm * 1e-3) / 1e9
        # PMC traffic of this kernel build, if a measurement of exactly these sources is committed (tools/measure_traffic.py)
        traffic, traffic_src, phys_read = None, None, None
        sha = kernel_source_sha()
        key = args.config + ("" if not args.mod else "/" + args.mod) + ("/no-eq" if args.no_eq else "")
        for tf in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_demod_traffic.json")), reverse=True):
            try:
                ent = json.load(open(os.path.join(ROOT, "profiles", tf))).get(key)
            except Exception:
                ent = None
            if isinstance(ent, dict) and ent.get("kernel_source_sha") == sha and ent.get("data_symbols_per_launch") == dsym:
                traffic = int(ent["hbm_bytes_per_launch_mean"])
                phys_read = ent["hbm_read_bytes_per_launch_mean"] / (dm * 1e-3) / 1e9
                traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel source, sha %s, kernel %s)" % (tf, sha, ent.get("kernel"))
                break
        copy_gbs = pat_gbs = power = None
        if world == 1 and not args.no_probes:
            # same-run context: this chip's float4 device-copy rate and the demod access pattern without arithmetic
            from ofdm_mi355x import _lib as ol
            d_rx0 = inputs[0][0]

            def _probe(mode):
                ts = []
                nb = d_eq.numel() * 4 if d_eq is not None else d_bits.numel()
                nb -= nb % 16
                dst = d_eq if d_eq is not None else d_bits
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ol.check(rxe.lib.ofdm_bandwidth_probe(device, ol.ptr(d_rx0), ol.ptr(dst), nb, mode, N * 8, cp * 8, (Kd * 8) & ~15, dsym if d_eq is not None else 0, stream))
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1))
                return float(np.median(ts)), nb
            cms, cnb = _probe(0)
            copy_gbs = 2 * cnb / cms / 1e6
            if d_eq is not None:
                pms, _ = _probe(1)
                pat_gbs = dsym * (L * 8 + Kd * 8) / pms / 1e6
            # package power and shader clock while the step runs back to back (untimed extra loop; rocm-smi children are not
            # started under a profiler's preload).  DESIGN.md section 4: this kernel sits at the package power limit.
            if not _profiled():
                try:
                    power = _power_sample(torch, step)
                except Exception:
                    power = None
        roof = dict(bound="hbm", kernel="rx_demod_kernel<%d>" % N, achieved=round(ach, 1), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=int(alg), kernel_ms=round(dm, 4), sync_kernel_ms=round(float(k_sync), 4),
                    kernel_source_sha=sha,
                    stream_read_only_frac=round(dsym * L * 8 / (dm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    physical_read_GBs=None if phys_read is None else round(phys_read, 1),
                    measured_copy_GBs=None if copy_gbs is None else round(copy_gbs, 1),
                    access_pattern_no_math_GBs=None if pat_gbs is None else round(pat_gbs, 1),
                    package_power_w_and_sclk_mhz_under_load=power)
        contention = None
        if world == 1 and args.contention_probe:
            # What the re-assembly would cost the HBM-bound demod at N = 8, measured on one GPU: per step every rank receives 7 x its
            # own packed bits into its receive buffer (HBM writes) and its own shard is read by 7 peers (HBM reads).  A device copy
            # of that size on a second stream reads and writes exactly that much while the step runs.
            nbytes = int(d_bits.numel()) * 7
            src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            side = torch.cuda.Stream(device=device)
            k3 = max(20, min(args.steps, 100))

            def timed(mode):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k3):
                    if mode:
                        with torch.cuda.stream(side):
                            if mode == 1:
                                dst.copy_(src, non_blocking=True)        # 7 shards written AND 7 shards read (peers reading ours)
                            else:
                                dst.fill_(1)                             # the inbound writes alone
                    step()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / k3 * 1e3
            timed(1)
            a0, a1, b1, a2, a3, b3 = timed(0), timed(1), timed(2), timed(0), timed(1), timed(2)
            wo, wi, ww = min(a0, a2), min(a1, a3), min(b1, b3)
            contention = dict(what="step with a concurrent device copy of 7 x the step's packed bits (all-gather traffic of N = 8 on this GPU's HBM)",
                              bytes_written_and_read_per_step=nbytes, steps=k3, ms_per_step_without=round(wo, 4), ms_per_step_with=round(wi, 4),
                              slowdown=round(wi / wo, 4), ms_per_step_with_writes_only=round(ww, 4), slowdown_writes_only=round(ww / wo, 4),
                              copy_GBs_if_alone=None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst.copy_(src, non_blocking=True)
            e1.record()
            e1.synchronize()
            contention["copy_GBs_if_alone"] = round(2 * nbytes * 10 / e0.elapsed_time(e1) / 1e6, 1)
            del src, dst
        cpu = None
        if world == 1 and not args.no_cpu:
            cpu = cpu_baseline(cfg, inputs[0][0], fl)
        out = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median": round(float(np.median(step_ms)), 4), "ms_per_step_min": round(float(step_ms.min()), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "frames_per_gpu": n_frames * batches, "symbols_per_gpu": n_frames * batches * n_sym,
                       "samples_per_step": samples_per_step, "outputs": ("bits" if d_eq is None else "equalised symbols + packed bits"),
                       "parallelism": "frame-shard x%d%s" % (world, " + RCCL all-gather of packed bits (%d sub-batches per step, gathers overlapped with later demod launches)" % n_chunks if world > 1 else ""),
                       # stream-equivalent: every input sample counted at 8 B, although CP samples and 59 of 60 sync symbols per
                       # frame are never fetched (BASELINE.md's "HBM-read roofline" definition); roofline.physical_read_GBs is fetched bytes
                       "stream_equivalent_read_fraction_of_8TBs": round(value * 1e6 * 8 / world / 8e12, 4),
                       "bit_error_rate_frame0": ber, "sync_search": sync_mode, "all_gather": gather_info,
                       "resident_plan_GB_per_gpu": round(need / 1e9, 1), "hbm_contention_probe": contention},
            "roofline": roof, "cpu_baseline": cpu,
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
