#!/usr/bin/env python3
"""Device-resident rate of the fused transmit kernel (bits -> grid -> IFFT -> CP -> power norm) and of the channel
kernel, against the HBM roofline: `python tools/tx_rate.py [nfft cp Kd mod frames]`.  Algorithmic bytes per symbol:
TX writes L*8 B and reads Kd*bps one-bit bytes per data symbol; the channel reads and writes L*8 B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd")]
import numpy as np, torch
import ofdm_mi355x as om
N, cp, Kd = (int(x) for x in (sys.argv[1:4] or (2048, 144, 1200)))
mod = sys.argv[4] if len(sys.argv) > 4 else "16QAM"
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 512
n_sym, L = 240, N + cp
bps = {"QPSK": 2, "16QAM": 4, "64QAM": 6}[mod]
tx = om.TxEngine(N, cp, N - 2, Kd, (1, 3), mod)
nb = tx.bits_per_frame(n_sym)
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))      # a real stream handle: 0 would mean "the engine's own stream"
stream = torch.cuda.current_stream().cuda_stream
for mode, name in ((om.BITS_UNPACKED, "one bit per byte"), (om.BITS_PACKED, "packed")):
    if mode == om.BITS_UNPACKED:
        bits = torch.randint(0, 2, (frames, nb), dtype=torch.uint8, device=dev)
    else:
        bits = torch.randint(0, 256, (frames, nb // 8), dtype=torch.uint8, device=dev)
    iq = torch.empty((frames, n_sym * L, 2), dtype=torch.float32, device=dev)
    out = torch.empty_like(iq)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(3):
        tx.modulate_frames(bits.data_ptr(), frames, n_sym, iq.data_ptr(), bits_mode=mode, stream=stream)
    torch.cuda.synchronize()
    reps = 10
    ev[0].record()
    for it in range(reps):
        tx.modulate_frames(bits.data_ptr(), frames, n_sym, iq.data_ptr(), bits_mode=mode, stream=stream)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    byts = frames * n_sym * L * 8 + bits.numel()
    print("tx_modulate %d-pt %s (%s): %.3f ms per %d frames -> %.0f GB/s algorithmic (%.2f of 8 TB/s), %.1f Gsamples/s"
          % (N, mod, name, ms, frames, byts / ms / 1e6, byts / ms / 1e6 / 8000, frames * n_sym * L / ms / 1e6))
taps = torch.zeros((1, 2), dtype=torch.float32, device=dev); taps[0, 0] = 1
for nv, name in ((0.0, "no noise"), (1e-3, "AWGN")):
    for it in range(2):
        tx.channel(iq.data_ptr(), frames, n_sym * L, n_sym * L, taps.data_ptr(), 1, out.data_ptr(), n_sym * L, n_sym * L, noise_var=nv, seed=1, stream=stream)
    torch.cuda.synchronize(); ev[0].record()
    for it in range(10):
        tx.channel(iq.data_ptr(), frames, n_sym * L, n_sym * L, taps.data_ptr(), 1, out.data_ptr(), n_sym * L, n_sym * L, noise_var=nv, seed=1, stream=stream)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    byts = 2 * frames * n_sym * L * 8
    print("channel (1 tap, %s): %.3f ms -> %.0f GB/s algorithmic (%.2f of 8 TB/s)" % (name, ms, byts / ms / 1e6, byts / ms / 1e6 / 8000))
