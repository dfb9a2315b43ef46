#!/usr/bin/env python3
"""One-off differential campaign of the fused transmit kernel (GPU): random numerology, prefix (odd, one sample, longer than N/2),
bin counts, [S, D] pattern, constellation, symbol and frame counts, packed and one-bit-per-byte input, against the fp64 oracle
(`tx_modulate`) at 1e-5, and the two bit layouts against each other bit for bit.  `python tools/fuzz_tx.py [seconds] [seed]`."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lte-gnu-radio-code_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import ofdm_mi355x as om
from oracle import ofdm_oracle as orc
from conftest import relerr

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
om.load()
t0 = time.time()
n_cases = n_fail = 0
sizes = {}
while time.time() - t0 < budget:
    logn = int(rng.integers(6, 13))
    N = 1 << logn
    u = rng.random()
    cp = 1 if u < 0.05 else int(rng.integers(N // 2, N)) if u < 0.12 else int(rng.integers(2, N // 4 + 1))
    Ks = N - 2 if rng.random() < 0.7 else int(rng.integers(N // 4, N // 2)) * 2
    Kd = min(int(rng.integers(1, Ks // 4 + 1)) * 4, Ks // 4 * 4)
    S, D = int(rng.integers(1, 3)), int(rng.integers(1, 5))
    mod = str(rng.choice(["BPSK", "QPSK", "16QAM", "64QAM"]))
    n_sym = (S + D) * int(rng.integers(1, 5))
    n_frames = int(rng.integers(1, max(2, min(40, int(3e5 // (n_sym * N))) + 1)))
    n_cases += 1
    sizes[N] = sizes.get(N, 0) + 1
    c = (N, cp, Ks, Kd, (S, D), mod, n_sym, n_frames)
    try:
        bps = orc.BITS_PER_SYMBOL[mod]
        tx = om.TxEngine(N, cp, Ks, Kd, (S, D), mod)
        nb = tx.bits_per_frame(n_sym)
        bits = rng.integers(0, 2, (n_frames, nb)).astype(np.uint8)
        L = N + cp
        d_bits = om.DeviceBuffer(bits.nbytes).upload(bits)
        d_iq = om.DeviceBuffer(n_frames * n_sym * L * 8)
        tx.modulate_frames(d_bits, n_frames, n_sym, d_iq, bits_mode=om.BITS_UNPACKED)
        iq = d_iq.download(np.complex64, n_frames * n_sym * L).reshape(n_frames, -1)
        for f in sorted({0, n_frames // 2, n_frames - 1}):
            ref = orc.tx_modulate(bits[f], N, cp, Ks, Kd, n_sym, synch_dat=(S, D), modulation=mod)
            e = relerr(iq[f], ref)
            assert e < 1e-5, "frame %d: %.3g" % (f, e)
        if nb % 8 == 0:
            pk = np.packbits(bits, axis=1)
            d_pk = om.DeviceBuffer(pk.nbytes).upload(pk)
            d_iq2 = om.DeviceBuffer(n_frames * n_sym * L * 8)
            tx.modulate_frames(d_pk, n_frames, n_sym, d_iq2, bits_mode=om.BITS_PACKED)
            assert np.array_equal(d_iq2.download(np.complex64, n_frames * n_sym * L), iq.ravel()), "packed != one bit per byte"
    except Exception as e:
        n_fail += 1
        print("FAIL case", c, "->", type(e).__name__, str(e)[:300], flush=True)
        if n_fail <= 2:
            traceback.print_exc()
print("fuzz_tx: %d cases (%s), %d failed" % (n_cases, sizes, n_fail))
sys.exit(1 if n_fail else 0)
