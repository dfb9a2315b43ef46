import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/lte-gnu-radio-code_amd") else os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd")]
import numpy as np, torch
import ofdm_mi355x as om
import RXOFDM, utsa_ofdm
N, cp, Kd, n_sym = 2048, 144, 1200, 240
L = N + cp
for name, mk, root in (("RXOFDM.synch_and_chan_est", lambda: RXOFDM.synch_and_chan_est(n_sym, N, cp, N - 2, [1, 3], Kd, 100, "/tmp/", "x", 0, 0), 37),
                       ("utsa_ofdm.SynchAndChanEst", lambda: utsa_ofdm.SynchAndChanEst(n_sym, N, cp, N - 2, [1, 3], Kd, 100, 0.7, "/tmp/", "x", 0, 0), 23)):
    txe = om.TxEngine(N, cp, N - 2, Kd, (1, 3), "QPSK", zc_root=root)
    bits = np.random.default_rng(0).integers(0, 2, txe.bits_per_frame(n_sym)).astype(np.uint8)
    d_b = om.DeviceBuffer(bits.nbytes).upload(bits)
    d_x = om.DeviceBuffer(n_sym * L * 8)
    txe.modulate_frames(d_b, 1, n_sym, d_x)
    iq = d_x.download(np.complex64, n_sym * L)
    blk = mk()
    out = np.zeros(len(iq), np.complex64)
    for _ in range(3):
        blk.work([iq], [out])
    t0 = time.perf_counter(); n = 20
    for _ in range(n):
        blk.work([iq], [out])
    dt = (time.perf_counter() - t0) / n
    print("%s work(): %.3f ms per %d-sample buffer -> %.1f Msamples/s; trials run in the last call: %s" % (name, dt * 1e3, len(iq), len(iq) / dt / 1e6, getattr(blk, "_rx").report.trials_run if hasattr(blk, "_rx") else "?"))
