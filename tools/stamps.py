#!/usr/bin/env python3
"""Where does a symbol's time go?  Runs the stamped diagnostic build of rx_demod_kernel<2048> (variant 9) and prints the
mean cycles per phase per symbol (per wave).  Shares, not absolute times: the stamps serialise (cdna_hip_programming.md 7)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools", "experiments")]
import numpy as np, torch
import explib                      # bench-only library (make -C lte-gnu-radio-code_amd/csrc exp)
import ofdm_mi355x as om
from ofdm_mi355x import _lib
import bench

cfg = dict(bench.CONFIGS["cfg2"]); n_frames = 2048
torch.cuda.set_device(0)
d_rx, _, _ = bench.build_inputs(torch, om, cfg, n_frames, 0, 1)
N, cp, Kd, n_sym = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["n_sym"]
fl = n_sym * (N + cp)
rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 30, 0.7, modulation=cfg["mod"])
rxe.reserve(n_frames); rxe.set_profiling(True)
nds = rxe.data_symbols_per_frame(fl)
d_eq = torch.empty((n_frames, nds, Kd, 2), dtype=torch.float32, device="cuda")
d_bits = torch.empty((n_frames, nds * Kd * 4 // 8), dtype=torch.uint8, device="cuda")
stamps = torch.zeros((65536 * 4, 8), dtype=torch.int32, device="cuda")
explib.set_stamp_buffer(rxe, stamps)
st = torch.cuda.current_stream().cuda_stream
for v in (0, 9, 9):
    explib.set_variant(rxe, v)
    rxe.demod_frames(d_rx, n_frames, fl, fl, d_eq, d_bits, om.BITS_PACKED, None, st)
    print("variant", v, "kernel ms", rxe.kernel_ms()[1])
s = stamps.cpu().numpy().astype(np.int64)
s = s[s.sum(1) > 0]
nsym_per_wave = n_frames * nds / (len(s) / 2)       # 2 waves per symbol slot; each wave sees every symbol of its slot
names = ["issue loads", "wait loads", "pass0+xchgA", "pass1+xchgB", "pass2+scatter", "list read+psum", "eq+demap+store", "end barrier"]
tot = s.sum()
print("waves", len(s), "symbols per wave %.1f" % nsym_per_wave)
for i, n in enumerate(names):
    print("%-16s %8.0f cycles/symbol  %5.1f %%" % (n, s[:, i].sum() / len(s) / nsym_per_wave, 100.0 * s[:, i].sum() / tot))
print("total %.0f cycles per symbol per wave" % (tot / len(s) / nsym_per_wave))
