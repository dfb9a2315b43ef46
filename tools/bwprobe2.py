#!/usr/bin/env python3
"""No-arithmetic probe of the demod access pattern at restricted occupancy (unused dynamic LDS limits workgroups per CU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd")]
import numpy as np, torch
import ofdm_mi355x as om
from ofdm_mi355x import _lib as ol
torch.cuda.set_device(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
lib = ol.load()
N, cp, Kd = 2048, 144, 1200
nsym = 786420
src = torch.empty(((nsym + 8) * (N + cp) * 2,), dtype=torch.float32, device="cuda").normal_()
dst = torch.empty((nsym * Kd * 2,), dtype=torch.float32, device="cuda")
def run(mode, kb, rd=True, wr=True):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ol.check(lib.ofdm_bandwidth_probe(0, ol.ptr(src), ol.ptr(dst), dst.numel() * 4, mode + 16 * kb, N * 8 if rd else 0, cp * 8 if rd else (N + cp) * 8, Kd * 8 if wr else 0, nsym, s.cuda_stream))
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for kb in (0, 20, 26, 32, 40, 53, 80, 159):
    wgs = 160 // kb if kb else 8
    ms = run(1, kb)
    print("pattern, %3d KiB LDS (<=%d WG/CU = %2d waves/CU): %.3f ms  %.0f GB/s algorithmic" % (kb, min(wgs, 8), min(wgs, 8) * 4, ms, nsym * ((N + cp) * 8 + Kd * 8) / ms / 1e6))
for kb in (0, 40, 80):
    ms = run(0, kb)
    print("copy, %3d KiB LDS: %.3f ms %.0f GB/s" % (kb, ms, 2 * dst.numel() * 4 / ms / 1e6))

for kb in (0, 40, 80):
    ms = run(1, kb, wr=False)
    print("read-only pattern,  %3d KiB LDS: %.3f ms  %.0f GB/s read (algorithmic incl. CP)" % (kb, ms, nsym * (N + cp) * 8 / ms / 1e6))
    ms = run(1, kb, rd=False)
    print("write-only pattern, %3d KiB LDS: %.3f ms  %.0f GB/s written" % (kb, ms, nsym * Kd * 8 / ms / 1e6))
