import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd")]
import numpy as np, torch
import ofdm_mi355x as om
import bench
cfg = dict(bench.CONFIGS[sys.argv[1]]); n_frames = int(sys.argv[2])
N, cp, Kd, mod, n_sym = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["mod"], cfg["n_sym"]
fl = n_sym * (N + cp)
torch.cuda.set_device(0)
d_rx, tx_bits = bench.build_inputs(torch, om, cfg, n_frames, 0, seed=99)
rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, cfg["snr_db"], 0.7, modulation=mod, device=0)
rxe.reserve(n_frames); rxe.set_max_trials(N + cp)
nds = rxe.data_symbols_per_frame(fl); nbytes = nds * Kd * bench.BPS[mod] // 8
st = torch.cuda.current_stream().cuda_stream
def run(src):
    bits = torch.empty((n_frames, nbytes), dtype=torch.uint8, device="cuda")
    tsr = torch.zeros((n_frames, 4), dtype=torch.int32, device="cuda")
    eq = torch.empty((n_frames, nds, Kd, 2), dtype=torch.float32, device="cuda") if len(sys.argv) > 3 else None
    rxe.demod_frames(src, n_frames, fl, fl, eq, bits, om.BITS_PACKED, tsr if os.environ.get('WITH_TSR') else None, st)
    torch.cuda.synchronize()
    return bits, tsr
b1, t1 = run(d_rx)
rev = torch.empty_like(d_rx)
for f0 in range(0, n_frames, 256):
    nf = min(256, n_frames - f0)
    rev[n_frames - f0 - nf:n_frames - f0] = torch.flip(d_rx[f0:f0 + nf], dims=[0])
torch.cuda.synchronize()
print("rev == flip(d_rx) rows:", [bool(torch.equal(rev[i], d_rx[n_frames - 1 - i])) for i in (0, 1, 2, 100, 599)])
b3, t3 = run(rev)
bad = [i for i in range(n_frames) if not torch.equal(b3[i], b1[n_frames - 1 - i])]
print("bad rows", len(bad), bad[:10], bad[-5:])
print("tsr rev bad rows", [t3[i].tolist() for i in bad[:4]], "partner", [t1[n_frames-1-i].tolist() for i in bad[:4]]); print("tsr rev first rows", t3[:4].tolist(), "orig last rows", t1[-4:].tolist())
b3b, t3b = run(rev)
bad2 = [i for i in range(n_frames) if not torch.equal(b3b[i], b1[n_frames - 1 - i])]
print("second run bad rows", len(bad2), bad2[:10])
