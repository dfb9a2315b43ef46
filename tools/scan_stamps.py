#!/usr/bin/env python3
"""Where does a block iteration of the screened sync search go?  Runs the stamped experiment build of rx_sync_scan_kernel
(s_memtime per phase, summed per workgroup) on cfg2 frames with a fixed lead.   usage: python tools/scan_stamps.py [lead ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools", "experiments")]
import numpy as np, torch
import explib
import ofdm_mi355x as om
import bench

leads = sys.argv[1:] or ["0", "145", "1104"]
cfg = dict(bench.CONFIGS["cfg2"]); n_frames = int(os.environ.get("KB_FRAMES", "4369"))
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())
N, cp, Kd, n_sym = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["n_sym"]
fl = n_sym * (N + cp)
names = ["loop top", "anchor trial", "edge loads", "scan+thr", "recurrence", "cand reduce", "exit", "finalize"]
for lead in leads:
    d_rx, _, _ = bench.build_inputs(torch, om, cfg, n_frames, 0, 1, lead=lead)
    rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 30, 0.7, modulation=cfg["mod"])
    rxe.reserve(n_frames); rxe.set_profiling(True); rxe.set_max_trials(2 * (N + cp))
    stamps = torch.zeros((n_frames, 8), dtype=torch.int32, device="cuda")
    explib.set_stamp_buffer(rxe, stamps)
    nds = rxe.data_symbols_per_frame(fl)
    d_bits = torch.empty((n_frames, nds * Kd * 4 // 8), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        rxe.demod_frames(d_rx, n_frames, fl, fl, None, d_bits, om.BITS_PACKED, None, st)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.int64)
    print("lead %s: sync kernel %.4f ms; mean shader-clock cycles per workgroup and phase (s_memtime):" % (lead, rxe.kernel_ms()[0]))
    for i, n in enumerate(names):
        print("   %-14s %10.0f" % (n, s[:, i].mean()))
    print("   total          %10.0f" % s.sum(1).mean())
    del d_rx, rxe
