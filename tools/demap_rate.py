#!/usr/bin/env python3
"""Device-resident rate of the standalone de-mapper (BitRecovery): hard bits only, and hard + both
max-log soft metrics (which needs the global sigma reduction first).  Algorithmic bytes per symbol: 8 B in, bps B out (hard);
soft adds 2 * bps * 4 B out and a second 8 B read for the sigma pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd")]
import numpy as np, torch
import ofdm_mi355x as om
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27
torch.cuda.set_stream(torch.cuda.Stream())
st = torch.cuda.current_stream().cuda_stream
rx = om.RxEngine(8, 64, 16, 62, (1, 3), 60, 100)
z = (torch.randn((n, 2), device="cuda") * 0.7)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for mod, bps in (("QPSK", 2), ("16QAM", 4), ("64QAM", 6)):
    hard = torch.empty(n * bps, dtype=torch.uint8, device="cuda")
    s0 = torch.empty(n * bps, dtype=torch.float32, device="cuda")
    s1 = torch.empty(n * bps, dtype=torch.float32, device="cuda")
    for name, args, byts in (("hard", (hard, None, None), n * (8 + bps)), ("hard + soft", (hard, s0, s1), n * (16 + bps + 8 * bps))):
        for _ in range(2):
            rx.demap(z, n, mod, *args, stream=st)
        torch.cuda.synchronize(); ev[0].record()
        for _ in range(5):
            rx.demap(z, n, mod, *args, stream=st)
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        print("demap %-6s %-11s: %.3f ms per %d symbols -> %.0f GB/s algorithmic (%.2f of 8 TB/s), %.1f Gsymbols/s"
              % (mod, name, ms, n, byts / ms / 1e6, byts / ms / 1e6 / 8000, n / ms / 1e6))
    del hard, s0, s1
