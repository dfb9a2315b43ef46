#!/usr/bin/env python3
"""Kernel-variant micro-benchmark: rx_demod_kernel<2048> variants, interleaved rounds in ONE process
(cdna_hip_programming.md rule 24).  usage: python tools/kbench.py [variants...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools", "experiments")]
import numpy as np, torch
import explib                      # bench-only library with the variants compiled in (make -C lte-gnu-radio-code_amd/csrc exp)
import ofdm_mi355x as om
import bench

variants = [int(x) for x in sys.argv[1:]] or [0]
cfg = dict(bench.CONFIGS[os.environ.get("KB_CONFIG", "cfg2")])
n_frames = int(os.environ.get("KB_FRAMES", "2048"))
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())   # a real stream handle (0 would mean the library's own stream)
d_rx, _, _ = bench.build_inputs(torch, om, cfg, n_frames, 0, 1)
N, cp, Kd, n_sym = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["n_sym"]
fl = n_sym * (N + cp)
rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 30, 0.7, modulation=cfg["mod"])
rxe.reserve(n_frames); rxe.set_profiling(True)
nds = rxe.data_symbols_per_frame(fl)
bps = bench.BPS[cfg["mod"]]
d_eq = torch.empty((n_frames, nds, Kd, 2), dtype=torch.float32, device="cuda")
d_bits = torch.empty((n_frames, nds * Kd * bps // 8), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
res = {v: [] for v in variants}
ref = None
# The chip runs this kernel at its package power limit: what ran just before shifts the clocks of what runs next.  Every variant
# therefore runs in sustained BLOCKS of KB_BURST back-to-back launches, of which only the second half is kept; blocks of the
# variants are interleaved KB_ROUNDS times (cdna_hip_programming.md rule 24).
burst = int(os.environ.get("KB_BURST", "12"))
for rnd in range(int(os.environ.get("KB_ROUNDS", "5"))):
    for v in variants:
        explib.set_variant(rxe, v)
        for b in range(burst):
            rxe.set_profiling(True)        # resets the library's event ring: kernel_ms() below is THIS call only (it averages the ring)
            rxe.demod_frames(d_rx, n_frames, fl, fl, None if os.environ.get('KB_NOEQ') == '1' else d_eq, d_bits, om.BITS_PACKED, None, st)
            ms = rxe.kernel_ms()[1]
            if b >= burst // 2:
                res[v].append(ms)
        if rnd == 0:
            h = (float(d_eq.double().sum().item()), int(d_bits.long().sum().item()))
            ref = ref or h
            if os.environ.get("KB_NOCHECK") != "1":
                assert h[1] == ref[1] and abs(h[0] - ref[0]) <= 1e-6 * abs(ref[0]) + 1e-3, "variant %d output differs: %r vs %r" % (v, h, ref)
alg = n_frames * nds * ((N + cp) * 8 + Kd * 8 + Kd * bps // 8)
for v in variants:
    r = np.array(res[v])
    print("variant %d: median %.3f ms  min %.3f ms  -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s), %.0f Gsamples/s"
          % (v, np.median(r), r.min(), alg / np.median(r) / 1e6, alg / np.median(r) / 1e6 / 80, n_frames * fl / np.median(r) / 1e6))
