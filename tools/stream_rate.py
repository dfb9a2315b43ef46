#!/usr/bin/env python3
"""PCIe-inclusive rate of the stream block (host buffers in, host buffers out): utsa_ofdm.SynchAndChanEst.work() on one
240-symbol buffer at 2048/144/1200.  This is the GNU Radio drop-in path; it is never bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd")]
import numpy as np, torch
import ofdm_mi355x as om
import utsa_ofdm
N, cp, Kd, n_sym = 2048, 144, 1200, 240
L = N + cp
txe = om.TxEngine(N, cp, N - 2, Kd, (1, 3), "QPSK")
bits = np.random.default_rng(0).integers(0, 2, txe.bits_per_frame(n_sym)).astype(np.uint8)
d_b = om.DeviceBuffer(bits.nbytes).upload(bits)
d_x = om.DeviceBuffer(n_sym * L * 8)
txe.modulate_frames(d_b, 1, n_sym, d_x)
iq = d_x.download(np.complex64, n_sym * L)
blk = utsa_ofdm.SynchAndChanEst(n_sym, N, cp, N - 2, [1, 3], Kd, 100, 0.7, "/tmp/", "x", 0, 0)
out = np.zeros(len(iq), np.complex64)
for _ in range(3):
    blk.work([iq], [out])
t0 = time.perf_counter(); n = 20
for _ in range(n):
    blk.work([iq], [out])
dt = (time.perf_counter() - t0) / n
print("stream block work(): %.3f ms per %d-sample buffer -> %.1f Msamples/s (H2D + sync search + demod + D2H + host packing)" % (dt * 1e3, len(iq), len(iq) / dt / 1e6))
for rows in (4, 32):                      # GNU Radio sized buffers: whole [S, D] patterns (anything else raises the reference's ValueError)
    chunk = rows * L
    b = iq[:chunk]; o = np.zeros(chunk, np.complex64)
    blk2 = utsa_ofdm.SynchAndChanEst(rows, N, cp, N - 2, [1, 3], Kd, 100, 0.7, "/tmp/", "x", 0, 0)
    blk2.work([b], [o]); t0 = time.perf_counter()
    for _ in range(20): blk2.work([b], [o])
    dt = (time.perf_counter() - t0) / 20
    print("  %6d-sample buffers (%d symbols): %.3f ms -> %.1f Msamples/s" % (chunk, rows, dt * 1e3, chunk / dt / 1e6))
