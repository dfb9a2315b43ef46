#!/usr/bin/env python3
"""What the host link gives for the stream block's buffer sizes: pageable vs pinned, H2D and D2H, per-call latency.
(tools/ only: context for DESIGN.md's PCIe-inclusive rate; uses torch for the copies.)"""
import time, numpy as np, torch
torch.cuda.init()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
for n_bytes in (70144, 561152, 4208640):
    host_pg = np.ones(n_bytes, np.uint8)
    host_pin = torch.ones(n_bytes, dtype=torch.uint8).pin_memory()
    dev = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    for name, src in (("pageable", torch.from_numpy(host_pg)), ("pinned", host_pin)):
        for direction in ("h2d", "d2h"):
            for _ in range(5):
                (dev.copy_(src, non_blocking=True) if direction == "h2d" else src.copy_(dev, non_blocking=True)); torch.cuda.synchronize()
            t0 = time.perf_counter(); reps = 50
            for _ in range(reps):
                (dev.copy_(src, non_blocking=True) if direction == "h2d" else src.copy_(dev, non_blocking=True)); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print("%8d B %-8s %s: %7.1f us  %6.2f GB/s" % (n_bytes, name, direction, dt * 1e6, n_bytes / dt / 1e9))
t0 = time.perf_counter()
for _ in range(200): torch.cuda.synchronize()
print("empty stream sync: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
a = np.ones(4208640, np.uint8); b = np.empty_like(a)
t0 = time.perf_counter()
for _ in range(50): np.copyto(b, a)
print("host memcpy 4.2 MB: %.1f us (%.1f GB/s)" % ((time.perf_counter() - t0) / 50 * 1e6, 4208640 / ((time.perf_counter() - t0) / 50) / 1e9))
