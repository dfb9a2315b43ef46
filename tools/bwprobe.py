#!/usr/bin/env python3
"""Achievable HBM rates on this chip: float4 copy, and the demod access pattern (2048/144/1200) without arithmetic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd")]
import torch
from ofdm_mi355x import _lib
lib = _lib.load()
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream
n_sym = 786420
L, N, Kd = 2192, 2048, 1200
src = torch.empty(n_sym * L * 8 // 4, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty(n_sym * Kd * 8 // 4, dtype=torch.float32, device="cuda")
def timed(fn, reps=6):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
nb = dst.numel() * 4
ms = timed(lambda: _lib.check(lib.ofdm_bandwidth_probe(0, _lib.ptr(src), _lib.ptr(dst), nb, 0, 0, 0, 0, 0, st)))
print("float4 copy   : %.3f ms  %.0f GB/s (read+write)" % (ms, 2 * nb / ms / 1e6))
ms = timed(lambda: _lib.check(lib.ofdm_bandwidth_probe(0, _lib.ptr(src), _lib.ptr(dst), 0, 1, N * 8, (L - N) * 8, Kd * 8, n_sym, st)))
alg = n_sym * (L * 8 + Kd * 8)
print("demod pattern : %.3f ms  %.0f GB/s algorithmic (L*8 read incl. CP + Kd*8 write), %.0f GB/s touched" % (ms, alg / ms / 1e6, n_sym * (N * 8 + Kd * 8) / ms / 1e6))
