#!/usr/bin/env python3
"""Package power and shader clock (rocm-smi) while one kernel runs back to back: the demod kernel, its no-arithmetic access-pattern
probe and the plain copy probe.  usage: python tools/power_probe.py   (on an MI355X box)"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd")]
import numpy as np, torch
import ofdm_mi355x as om
from ofdm_mi355x import _lib as ol
import bench

torch.cuda.set_device(0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
cfg = dict(bench.CONFIGS["cfg2"]); n_frames = cfg["frames"]
d_rx, _, _ = bench.build_inputs(torch, om, cfg, n_frames, 0, 1)
N, cp, Kd, n_sym = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["n_sym"]
fl = n_sym * (N + cp)
rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 30, 0.7, modulation=cfg["mod"])
rxe.reserve(n_frames); rxe.set_max_trials(N + cp)
nds = rxe.data_symbols_per_frame(fl)
d_eq = torch.empty((n_frames, nds, Kd, 2), dtype=torch.float32, device="cuda")
d_bits = torch.empty((n_frames, nds * Kd * 4 // 8), dtype=torch.uint8, device="cuda")
st = ts.cuda_stream
lib = ol.load()
nb = d_eq.numel() * 4
dsym = n_frames * nds

def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Package Power \(W\): ([0-9.]+)", out)
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else float("nan"), int(c.group(1)) if c else -1)

def measure(name, launch, seconds=8.0):
    samples, stop = [], [False]
    def sampler():
        time.sleep(2.5)
        while not stop[0]:
            samples.append(smi()); time.sleep(0.5)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(20):
            launch()
        n += 20
        torch.cuda.synchronize()
    e1.record(); e1.synchronize()
    stop[0] = True; th.join()
    ms = e0.elapsed_time(e1) / n
    pw = np.array([s[0] for s in samples]); ck = np.array([s[1] for s in samples])
    print("%-28s %.3f ms/launch   package power %.0f W (max %.0f)   sclk %.0f MHz (min %d, max %d)   [%d samples]"
          % (name, ms, np.median(pw), pw.max(), np.median(ck), ck.min(), ck.max(), len(samples)), flush=True)

print("idle: %.0f W, sclk %d MHz" % smi())
measure("demod kernel (sync+demod)", lambda: rxe.demod_frames(d_rx, n_frames, fl, fl, d_eq, d_bits, om.BITS_PACKED, None, st))
measure("demod, no eq stores", lambda: rxe.demod_frames(d_rx, n_frames, fl, fl, None, d_bits, om.BITS_PACKED, None, st))
measure("no-arithmetic pattern probe", lambda: ol.check(lib.ofdm_bandwidth_probe(0, ol.ptr(d_rx), ol.ptr(d_eq), nb, 1, N * 8, cp * 8, Kd * 8, dsym, st)))
measure("plain copy probe", lambda: ol.check(lib.ofdm_bandwidth_probe(0, ol.ptr(d_rx), ol.ptr(d_eq), nb, 0, 0, 0, 0, 0, st)))
print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout.strip().splitlines()[-3:])
