"""CPU oracle for the OFDM TX/RX hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a NumPy (complex128) restatement of the algorithm of the reference
repository's hot path (tayloreisman16/LTE-GNU-Radio-Code).  It is the checker the
HIP kernels are compared against.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product path
(``lte-gnu-radio-code_amd/``) never does and fails loudly without its HIP library.

Pinning: every function below is checked in ``tests/test_oracle_golden.py`` against
(i) the reference's own data fixtures (statically extracted, see
``tests/golden/gen_golden.py``) and (ii) outputs of the reference Python itself, run
once in the build container by the same script and committed under ``tests/golden``.
16-QAM / 64-QAM mapping is an extension the reference does not have
(``OFDM.py:12-15`` knows BPSK/QPSK only): parity UNPINNED for those two maps.

Citations use  G/ = /root/reference/GNU-Radio-Repositories/ :
  RX   G/gr-utsa_ofdm/python/SynchAndChanEst.py            (authoritative)
  RXc  G/gr-RXOFDM/python/synch_and_chan_est.py            (compat constants)
  TX   G/LEGACY/gr-ofdm-rx/python/txrx_mod/MultiAntennaSystem.py
  ZC   G/LEGACY/gr-ofdm-rx/python/txrx_mod/SynchSignal.py
  BR   G/LEGACY/gr-ofdm-rx/python/BitRecovery.py
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- bins / ZC


def bins_p(num_bins: int, nfft: int) -> np.ndarray:
    """Used-bin list, negative frequencies first, DC skipped (RX:38-41,66-70; ZC:13-14)."""
    h = int(num_bins / 2)
    neg = list(range(-h, 0, 1))
    pos = list(range(1, h + 1, 1))
    return (np.array(neg + pos, dtype=np.int64) + nfft) % nfft


def zadoff_chu(mm: int, root: int = 23) -> np.ndarray:
    """ZC sequence of length MM (RX:52-59, ZC:23-30; root 37 form RXc:54-64)."""
    x0 = np.arange(0, int(mm), dtype=np.float64)
    x1 = np.arange(1, int(mm) + 1, dtype=np.float64)
    if mm % 2 == 0:
        return np.exp(-1j * (2 * np.pi / mm) * root * (x0 ** 2 / 2))
    return np.exp(-1j * (2 * np.pi / mm) * root * (x0 * x1) / 2)


# --------------------------------------------------------------------------- bit <-> symbol maps

_QPSK_POINTS = np.exp(1j * 2 * np.pi / 8 * np.array([1.0, -1.0, 3.0, 5.0]))  # TX:171-178, BR:45-52

BITS_PER_SYMBOL = {"BPSK": 1, "QPSK": 2, "16QAM": 4, "64QAM": 6}


def map_bits(bits: np.ndarray, modulation: str) -> np.ndarray:
    """bits (flat, 0/1) -> complex symbols, MSB-first per symbol.

    BPSK ``2b-1`` (TX:156-157); QPSK index ``2*b0+b1`` -> exp(j*2pi/8*{1,-1,3,5}) (TX:159-178).
    16QAM/64QAM: 3GPP TS 36.211 section 7.1 Gray maps, unit average energy (extension, unpinned).
    """
    bits = np.asarray(bits).astype(np.int64).ravel()
    bps = BITS_PER_SYMBOL[modulation]
    b = bits.reshape(-1, bps)
    if modulation == "BPSK":
        return (2 * b[:, 0] - 1).astype(np.complex128)
    if modulation == "QPSK":
        return _QPSK_POINTS[2 * b[:, 0] + b[:, 1]]
    s = 1 - 2 * b  # 0 -> +1, 1 -> -1
    if modulation == "16QAM":
        i = s[:, 0] * (2 - s[:, 2])
        q = s[:, 1] * (2 - s[:, 3])
        return (i + 1j * q) / np.sqrt(10.0)
    if modulation == "64QAM":
        i = s[:, 0] * (4 - s[:, 2] * (2 - s[:, 4]))
        q = s[:, 1] * (4 - s[:, 3] * (2 - s[:, 5]))
        return (i + 1j * q) / np.sqrt(42.0)
    raise ValueError(modulation)


SQRT2_F32 = np.float32(1.41421354)   # largest float32 below sqrt(2): decision edge of BitRecovery, see below


def demap_hard(z: np.ndarray, modulation: str) -> np.ndarray:
    """Hard decision, output bit order [b0, b1, ...] per symbol (z is taken as complex64).

    QPSK follows BitRecovery.work literally (BR:105-157), whose hard bit is
    ``int(0.5*(sign(llr1-llr0)+1))`` with llr = -0.5*f*|e| / -0.5*f*(K-|e|), K = 1.414213562373095:
        b0 = 1  iff  -sqrt2 <= Re z < 0   or   Re z > sqrt2        (same for b1 with Im z)
    i.e. the sign rule PLUS the reference's outlier flip beyond |x| > sqrt(2) (its "far" metric
    K-|e| goes negative there).  For float32 inputs the edges are exactly +-SQRT2_F32.  This closed
    form equals ``bit_recovery(z)[0]`` whenever neither coordinate is exactly zero; a symbol ON an axis is a
    tie of the reference's nearest-point search, decided by the last bit of its fp64 arithmetic: those
    symbols are handed to ``bit_recovery`` (the literal restatement) here as well.
    16QAM / 64QAM: 3GPP TS 36.211 7.1 decision regions (extension, unpinned).
    """
    z = np.asarray(z).astype(np.complex64).ravel()
    re, im = z.real, z.imag
    if modulation == "BPSK":
        return (re > 0).astype(np.uint8)
    if modulation == "QPSK":
        t = SQRT2_F32
        b0 = ((re < 0) & (re >= -t)) | (re > t)
        b1 = ((im < 0) & (im >= -t)) | (im > t)
        out = np.stack([b0, b1], axis=1).astype(np.uint8)
        tie = np.nonzero((re == 0) | (im == 0))[0]
        if tie.size:        # the hard bit of a tie does not depend on the other symbols of the buffer (sigma only scales it)
            out[tie] = bit_recovery(np.concatenate([z[tie], np.array([1 + 1j], np.complex64)]))[0][:-2].reshape(-1, 2)
        return out.ravel()
    if modulation == "16QAM":
        t = np.float32(2.0 / np.sqrt(10.0))
        return np.stack([re < 0, im < 0, np.abs(re) > t, np.abs(im) > t], axis=1).astype(np.uint8).ravel()
    if modulation == "64QAM":
        a, c = np.float32(4.0 / np.sqrt(42.0)), np.float32(2.0 / np.sqrt(42.0))
        return np.stack([re < 0, im < 0, np.abs(re) > a, np.abs(im) > a,
                         np.abs(np.abs(re) - a) > c, np.abs(np.abs(im) - a) > c],
                        axis=1).astype(np.uint8).ravel()
    raise ValueError(modulation)


def _cabs_avx512(a: float, b: float) -> float:
    """|a + jb| as NumPy's AVX-512 kernel forms it (numpy/_core/src/umath/loops_*: avx512_cabsolute): every step one
    correctly rounded IEEE operation."""
    from fractions import Fraction
    a, b = abs(float(a)), abs(float(b))
    mx, mn = max(a, b), min(b, a)
    if not np.isfinite(mx) or mx == 0.0:
        return mx
    r = mn / mx
    return float(np.sqrt(np.float64(float(Fraction(r) * Fraction(r) + 1)))) * mx


def bit_recovery(z: np.ndarray):
    """Literal restatement of BitRecovery.work (BR:66-157) for QPSK.

    Returns (hardbit int[2n], softbit0 float[2n], softbit1 float[2n]).
    """
    z = np.asarray(z).astype(np.complex64).astype(np.complex128).ravel()
    n = len(z)
    cdat = _QPSK_POINTS
    zobs = z[:, None] - cdat[None, :]                       # BR:82-86
    dist = np.abs(zobs)
    # A symbol with an exactly-zero coordinate is, in exact arithmetic, equidistant from two (four) points: the reference's
    # arg-min there is decided by the last bit of np.abs.  The recorded reference run (tests/golden/ref_bitrecovery.npz) used
    # NumPy's AVX-512 complex-abs kernel, |a+jb| = max*sqrt(fma(r,r,1)), r = min/max; hosts without AVX-512 take libm's hypot and
    # can differ in that last bit.  The tie rows are therefore restated with that formula explicitly (exact fma by rationals).
    for i in np.nonzero((z.real == 0) | (z.imag == 0))[0]:
        dist[i] = [_cabs_avx512(v.real, v.imag) for v in zobs[i]]
    dminind = np.argmin(dist, axis=1)                       # BR:87
    dmin = np.min(dist, axis=1)                             # BR:88
    ez = z - cdat[dminind]                                  # BR:93-98
    sigma = 0.7071067811865476 * np.mean(np.abs(dmin))      # BR:102
    dfact = 1.0 / (sigma * sigma)                           # BR:103
    K = 1.414213562373095                                   # BR:57
    llrp0 = np.zeros(2 * n)
    llrp1 = np.zeros(2 * n)
    near_r = -0.5 * dfact * np.abs(ez.real)
    far_r = -0.5 * dfact * (K - np.abs(ez.real))
    near_i = -0.5 * dfact * np.abs(ez.imag)
    far_i = -0.5 * dfact * (K - np.abs(ez.imag))
    re, im = z.real, z.imag
    # quadrant tests in the reference's order (++, -+, --, +-), first match wins (BR:106-125)
    q1 = (re >= 0) & (im >= 0)
    q2 = ~q1 & (re <= 0) & (im >= 0)
    q3 = ~q1 & ~q2 & (re <= 0) & (im <= 0)
    q4 = ~q1 & ~q2 & ~q3 & (re >= 0) & (im <= 0)
    re_pos = q1 | q4          # real-bit decided "0" side
    im_pos = q1 | q2
    anyq = q1 | q2 | q3 | q4  # NaNs match nothing -> stay 0
    llrp0[0::2] = np.where(anyq, np.where(re_pos, near_r, far_r), 0.0)
    llrp1[0::2] = np.where(anyq, np.where(re_pos, far_r, near_r), 0.0)
    llrp0[1::2] = np.where(anyq, np.where(im_pos, near_i, far_i), 0.0)
    llrp1[1::2] = np.where(anyq, np.where(im_pos, far_i, near_i), 0.0)
    hard = (0.5 * (np.sign(llrp1 - llrp0) + 1.0)).astype(int)   # BR:155-156
    return hard, llrp0, llrp1


def qam_levels(modulation: str):
    """Per-axis PAM levels of the TS 36.211 7.1 maps used by map_bits, with the bit labels of the axis.

    Returns (levels float64[M], labels uint8[M, bps/2]); axis bit j of the I axis is symbol bit 2j, of the Q axis 2j+1."""
    if modulation == "16QAM":
        m = np.array([-3, -1, 1, 3])
        lab = np.stack([m < 0, np.abs(m) == 3], axis=1)
        return m / np.sqrt(10.0), lab.astype(np.uint8)
    if modulation == "64QAM":
        m = np.array([-7, -5, -3, -1, 1, 3, 5, 7])
        lab = np.stack([m < 0, np.abs(m) > 4, (np.abs(m) == 1) | (np.abs(m) == 7)], axis=1)
        return m / np.sqrt(42.0), lab.astype(np.uint8)
    raise ValueError(modulation)


def soft_demap_qam(z: np.ndarray, modulation: str):
    """16/64-QAM extension of BitRecovery's soft metric.  PARITY UNPINNED: the reference implements QPSK
    only (BR:45-52); this keeps its structure -- nearest-point search, sigma = 0.7071*mean(dmin) over the call's buffer
    (BR:87-88,102), per-bit metrics -0.5/sigma^2 * (linear distance) for the hypotheses bit=0 / bit=1 (BR:106-125), hard bit
    int(0.5*(sign(m1-m0)+1)) (BR:155-156) -- with the distance taken, per axis, to the nearest PAM level carrying that bit
    value (the max-log rule; for QPSK inliers it reduces to the reference's |e| / K-|e| pair).

    Returns (hardbit int[bps*n], softbit0 float[bps*n], softbit1 float[bps*n]), bit order b0..b(bps-1) per symbol."""
    z = np.asarray(z).astype(np.complex64).astype(np.complex128).ravel()
    n = len(z)
    lv, lab = qam_levels(modulation)
    nb = lab.shape[1]
    bps = 2 * nb
    d = [np.abs(x[:, None] - lv[None, :]) for x in (z.real, z.imag)]         # [axis][n, M]
    e = [np.min(dd, axis=1) for dd in d]
    dmin = np.hypot(e[0], e[1])                                               # distance to the nearest point (BR:88)
    sigma = 0.7071067811865476 * np.mean(dmin)                                # BR:102
    hf = -0.5 / (sigma * sigma)                                               # BR:103
    s0 = np.zeros((n, bps))
    s1 = np.zeros((n, bps))
    for axis in (0, 1):
        for j in range(nb):
            one = lab[:, j] == 1
            s0[:, 2 * j + axis] = hf * np.min(d[axis][:, ~one], axis=1)
            s1[:, 2 * j + axis] = hf * np.min(d[axis][:, one], axis=1)
    s0, s1 = s0.ravel(), s1.ravel()
    hard = (0.5 * (np.sign(s1 - s0) + 1.0)).astype(int)
    return hard, s0, s1


# --------------------------------------------------------------------------- TX


def tx_grid(bits: np.ndarray, nfft: int, num_synch_bins: int, num_data_bins: int,
            n_sym: int, synch_dat=(1, 3), modulation: str = "QPSK", zc_root: int = 23, zc_segments: bool = False,
            zc_parity_of_bins: bool = False, data_symbols=None) -> np.ndarray:
    """Resource grid (n_sym, nfft): ZC on sync symbols, mapped data on data symbols.

    TX:135-183.  ``synch_state`` never advances (TX:146) so every sync symbol carries
    ``zc[0:Ks]``.  Bits are consumed data-symbol-major, then bin-major in bins_p order.
    """
    S, D = int(synch_dat[0]), int(synch_dat[1])
    bps = BITS_PER_SYMBOL[modulation]
    sb = bins_p(num_synch_bins, nfft)
    db = bins_p(num_data_bins, nfft)
    zc = zadoff_chu(S * num_synch_bins, zc_root)
    if zc_parity_of_bins:      # gr-RXOFDM / SynchEstAndFO generation: the ZC form follows the parity of Ks, not of S*Ks
        mm = S * num_synch_bins
        t0 = np.arange(mm, dtype=np.float64)
        zc = np.exp((-1j * (2 * np.pi / mm) * zc_root / 2.0) * (t0 * t0 if num_synch_bins % 2 == 0 else t0 * (t0 + 1)))
    grid = np.zeros((n_sym, nfft), dtype=np.complex128)
    bits = np.asarray(bits).ravel()
    nd = 0
    for s in range(n_sym):
        if s % (S + D) < S:
            # reference TX: synch_state never advances -> segment 0 every time (TX:143-147); zc_segments=True sends
            # the segment the receivers correlate against (needed for synch_dat[0] > 1)
            seg = (s % (S + D)) if zc_segments else 0
            grid[s, sb] = zc[seg * num_synch_bins:(seg + 1) * num_synch_bins]
        else:
            if data_symbols is not None:      # pre-mapped bin values [n_data][Kd] (e.g. DSSS-spread symbols)
                grid[s, db] = np.asarray(data_symbols)[nd]
            else:
                chunk = bits[nd * num_data_bins * bps:(nd + 1) * num_data_bins * bps]
                grid[s, db] = map_bits(chunk, modulation)
            nd += 1
    return grid


def tx_cp_norm(x: np.ndarray, cp_len: int) -> np.ndarray:
    """One symbol's time samples -> CP-extended, power-normalised symbol (TX:200-218)."""
    t = np.concatenate((x[-cp_len:], x)) if cp_len > 0 else x.copy()         # TX:200-201
    e = abs(np.dot(t, np.conj(t)))                                           # TX:202
    if e > 1e-30:                                                            # TX:204
        t = t * np.sqrt(len(t) / e)                                          # TX:205-207
    p = np.var(t)                                                            # TX:213
    return t * (1 / np.sqrt(p))                                              # TX:218


def tx_symbol_synth(grid: np.ndarray, cp_len: int) -> np.ndarray:
    """IFFT + CP + power normalisation, symbol by symbol (TX:189-218). Returns flat IQ."""
    n_sym, nfft = grid.shape
    L = nfft + cp_len
    out = np.zeros(n_sym * L, dtype=np.complex128)
    for s in range(n_sym):
        out[s * L:(s + 1) * L] = tx_cp_norm(np.fft.ifft(grid[s], nfft), cp_len)     # TX:199
    return out


# ---- the decomposed transmitter (block names only in LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc:701-975; no reference code).
# Each stage is the matching slice of TX:135-218; chained with no pilots they ARE tx_modulate (asserted in the tests).
# Pilots and the counter-based bit source are this project's own definitions: parity unpinned.

def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on uint32 arrays; returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & 0xFFFFFFFF for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


def random_bits(seed: int, offset: int, n: int) -> np.ndarray:
    """Bits [offset, offset+n) of the random_bit_source stream: bit k = bit (k%32) of word (k/32)%4 of
    Philox4x32-10(counter = k/128, key = seed)  (include/ofdm_mi355x.h: ofdm_tx_random_bits)."""
    k = np.arange(offset, offset + n, dtype=np.uint64)
    blk = k >> np.uint64(7)
    ub, inv = np.unique(blk, return_inverse=True)
    w = np.stack(philox4x32_10(ub & np.uint64(0xFFFFFFFF), ub >> np.uint64(32), 0 * ub, 0 * ub, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    j = (k & np.uint64(127)).astype(np.int64)
    words = w[j >> 5, inv]
    return ((words >> (j & 31).astype(np.uint64)) & np.uint64(1)).astype(np.uint8)


def tx_stage_grid(symbols: np.ndarray, nfft: int, num_data_bins: int, pilot_locations=(), pilot_value=1.0 + 0j) -> np.ndarray:
    """OFDM_Modulation: rows of Kd symbols -> rows of nfft bins.  Occupied bins = bins_p(Kd + n_pilots) (TX:135-139), the
    signed offsets `pilot_locations` carry `pilot_value`, data fill the remaining occupied bins in list order (TX:182-183)."""
    sym = np.asarray(symbols).reshape(-1, num_data_bins)
    K = num_data_bins + len(pilot_locations)
    occ = bins_p(K, nfft)
    pil = {(int(p) + nfft) % nfft for p in pilot_locations}
    assert len(pil) == len(pilot_locations) and all(b in set(occ.tolist()) for b in pil)
    data_bins = np.array([b for b in occ if b not in pil], dtype=np.int64)
    grid = np.zeros((sym.shape[0], nfft), dtype=np.complex128)
    grid[:, data_bins] = sym
    if pil:
        grid[:, sorted(pil)] = pilot_value
    return grid


def tx_stage_ifft(grid: np.ndarray) -> np.ndarray:
    """IFFT: one numpy.fft.ifft per row (TX:199)."""
    return np.fft.ifft(np.asarray(grid), axis=1)


def tx_stage_cp(time_rows: np.ndarray, cp_len: int) -> np.ndarray:
    """CyclicPrefix: CP + power normalisation per row (TX:200-218)."""
    return np.stack([tx_cp_norm(r, cp_len) for r in np.asarray(time_rows)])


def tx_stage_mux(data_rows: np.ndarray, nfft: int, cp_len: int, prime_no: int, synch_every: int, synch_length: int) -> np.ndarray:
    """SynchDataMux: one ZC sync symbol (root prime_no on bins_p(synch_length), ZC:13-30, through the same IFFT + CP +
    normalisation) in front of every `synch_every` data symbols; a trailing partial group keeps its sync symbol."""
    g = np.zeros(nfft, dtype=np.complex128)
    g[bins_p(synch_length, nfft)] = zadoff_chu(synch_length, prime_no)
    sync = tx_cp_norm(np.fft.ifft(g, nfft), cp_len)
    rows = []
    for i, r in enumerate(np.asarray(data_rows)):
        if i % synch_every == 0:
            rows.append(sync)
        rows.append(r)
    return np.stack(rows) if rows else np.zeros((0, nfft + cp_len), dtype=np.complex128)


def tx_modulate(bits, nfft, cp_len, num_synch_bins, num_data_bins, n_sym,
                synch_dat=(1, 3), modulation="QPSK", zc_root=23, zc_segments=False, zc_parity_of_bins=False,
                data_symbols=None) -> np.ndarray:
    """bits -> time-domain IQ (a1+a2+a3)."""
    return tx_symbol_synth(
        tx_grid(bits, nfft, num_synch_bins, num_data_bins, n_sym, synch_dat, modulation, zc_root, zc_segments,
                zc_parity_of_bins, data_symbols), cp_len)


def dsss_spread(symbols: np.ndarray, dsss: int, num_data_bins: int) -> np.ndarray:
    """Transmit-side counterpart of DS:391-399 (the reference has no transmitter for it): each symbol of a row occupies DSSS
    consecutive listed bins, multiplied by the spreading code; bins past floor(Kd/DSSS)*DSSS stay empty."""
    symbols = np.atleast_2d(np.asarray(symbols))
    n = num_data_bins // dsss
    out = np.zeros((symbols.shape[0], num_data_bins), dtype=np.complex128)
    out[:, :n * dsss] = (symbols[:, :n, None] * spreading_code(dsss)[None, None, :]).reshape(symbols.shape[0], n * dsss)
    return out


REF_TAPS = np.array([0.3977, 0.7954 - 0.3977j, -0.1988, 0.0994, -0.0398])  # TX:64


def channel_apply(tx: np.ndarray, taps: np.ndarray, nfft: int) -> np.ndarray:
    """y = convolve(tx, taps/||taps|| zero-padded to nfft taps); length +nfft-1 (TX:79-94,221-231)."""
    h = np.zeros(nfft, dtype=np.complex128)
    taps = np.asarray(taps, dtype=np.complex128)
    h[0:len(taps)] = taps / np.linalg.norm(taps)
    return np.convolve(np.asarray(tx).ravel(), h)


def awgn_noise_var(tx: np.ndarray, nfft: int, cp_len: int, num_data_bins: int, bps: int,
                   snr_db: float, snr_type: str = "Digital") -> float:
    """Noise variance used by additive_noise (TX:235-248)."""
    sig_pow = np.var(tx)
    if snr_type == "Digital":
        return (1.0 / (num_data_bins * bps)) * (nfft + cp_len) * sig_pow * 10 ** (-snr_db / 10)
    return sig_pow * 10 ** (-snr_db / 10)


# --------------------------------------------------------------------------- RX


class RxOracle:
    """fp64 restatement of SynchAndChanEst (RX:17-262) with the reference's call-to-call state.

    ``compat='utsa'``  : gr-utsa_ofdm semantics (root 23, stride 1, gate param, SNR_lin law).
    ``compat='rxofdm'``: gr-RXOFDM constants (root 37, stride cp-1, gate 0.4, linear SNR)
                         applied to the same single-sync control flow (the gr-RXOFDM block itself
                         crashes under Python 3, RXc:194,253; only its constants are kept).
    """

    def __init__(self, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins,
                 snr, scale_factor_gate=0.7, compat="utsa", force_fp64=False):
        self.force_fp64 = bool(force_fp64)
        self.num_ofdm_symb = int(num_ofdm_symb)
        self.nfft = int(nfft)
        self.cp_len = int(cp_len)
        self.num_synch_bins = int(num_synch_bins)
        self.synch_dat = [int(synch_dat[0]), int(synch_dat[1])]
        self.num_data_bins = int(num_data_bins)
        self.synch_bins_used_P = bins_p(num_synch_bins, nfft)              # RX:38-41
        self.bins_used_P = bins_p(num_data_bins, nfft)                     # RX:66-70
        self.L_synch = len(self.synch_bins_used_P)
        self.M = [self.synch_dat[0], self.num_synch_bins]                  # RX:48
        self.MM = int(np.prod(self.M))                                     # RX:49
        self.compat = compat
        if compat == "utsa":
            self.p = 23                                                    # RX:52
            self.stride_val = 1                                            # RX:77
            self.gate = float(scale_factor_gate)                           # RX:166
            self.SNR = snr                                                 # RX:98
            self.SNR_lin = 10 ** (snr / 20)                                # RX:99 (sic: /20)
            self.snr_ls = self.SNR_lin                                     # RX:180
            self.snr_eqsync = self.SNR                                     # RX:214 (plain snr)
            self.snr_data = self.SNR_lin                                   # RX:245
        elif compat == "rxofdm":
            self.p = 37                                                    # RXc:54
            self.stride_val = self.cp_len - 1                              # RXc:81
            self.gate = 0.4                                                # RXc:170
            self.SNR = snr                                                 # RXc:102
            self.SNR_lin = snr
            self.snr_ls = snr                                              # RXc:184
            self.snr_eqsync = snr                                          # RXc:217
            self.snr_data = snr                                            # RXc:247
        else:
            raise ValueError(compat)
        self.zadoff_chu = zadoff_chu(self.MM, self.p)                      # RX:54-59
        self.start_samp = self.cp_len                                      # RX:78
        self.rx_b_len = self.nfft + self.cp_len                            # RX:79
        # del_mat_exp[d, i] = exp(j 2pi d k_i / N), tiled S times (RX:74-75)
        self.del_mat_exp = np.tile(
            np.exp((1j * (2.0 * np.pi / self.nfft)) *
                   np.outer(np.arange(self.cp_len + 1), self.synch_bins_used_P)), (1, self.M[0]))
        self.corr_obs = -1                                                 # RX:72
        self.count = 0                                                     # RX:100
        self.time_synch_ref = np.zeros(3)                                  # RX:83
        n = self.num_ofdm_symb
        self.est_chan_time = np.zeros((n, self.nfft), dtype=complex)       # RX:84
        self.est_synch_freq = np.zeros((n, self.MM), dtype=complex)        # RX:85
        self.est_chan_freq_P = np.zeros((n, self.nfft), dtype=complex)     # RX:87
        self.est_data_freq = np.zeros((n, self.num_data_bins), dtype=complex)  # RX:88
        self.eq_gain = None
        self.del_mat = None
        self.trials_run = 0

    # -- a7: one sync trial ------------------------------------------------------------
    def sync_trial(self, in0: np.ndarray, P: int):
        """Returns (del_mat[cp+1], y[MM]) for trial P (RX:145-161), O((cp+1)*MM) form."""
        S, N, L = self.M[0], self.nfft, self.rx_b_len
        y = np.zeros(self.MM, dtype=complex)
        for LL in range(S):
            a = L * LL + P * self.stride_val + self.start_samp             # RX:146
            w = in0[a:a + N].astype(np.complex128)                         # RX:148
            Y = np.fft.fft(w, N)                                           # RX:152
            y[LL * self.num_synch_bins:(LL + 1) * self.num_synch_bins] = Y[self.synch_bins_used_P]  # RX:153-154
        p_est = np.sqrt(len(y) / np.sum(y * np.conj(y)))                   # RX:157
        y = p_est * y                                                      # RX:159
        # del_mat_exp @ diag(y) @ conj(zc) == del_mat_exp @ (y * conj(zc))   (RX:160-161)
        del_mat = self.del_mat_exp @ (y * np.conj(self.zadoff_chu))
        return del_mat, y

    # -- a8: LS estimate + gains -------------------------------------------------------
    def _accept(self, P, dmax_ind, dmax_val, y):
        self.corr_obs += 1                                                 # RX:171
        self.time_synch_ref[0] = P * self.stride_val + self.start_samp    # RX:173
        self.time_synch_ref[1] = dmax_ind                                  # RX:174
        self.time_synch_ref[2] = int(dmax_val)                             # RX:175
        data_recov = self.del_mat_exp[dmax_ind] * y                        # RX:177-178
        tmp_v1 = data_recov * np.conj(self.zadoff_chu) / (1.0 + 1.0 / self.snr_ls)   # RX:180-181
        chan_est = np.sum(tmp_v1.reshape(self.M[0], self.L_synch), axis=0) / float(self.M[0])  # RX:183-184
        chan_est1 = np.zeros(self.nfft, dtype=complex)
        chan_est1[self.synch_bins_used_P] = chan_est                       # RX:186-187
        self.est_chan_freq_P[self.corr_obs] = chan_est1                    # RX:188
        self.est_chan_time[self.corr_obs] = np.fft.ifft(chan_est1, self.nfft)  # RX:202,212
        chan_mag = chan_est * np.conj(chan_est)                            # RX:213
        self.eq_gain = np.conj(chan_est) / (1.0 / self.snr_eqsync + chan_mag)   # RX:214-216
        self.est_synch_freq[self.corr_obs] = np.tile(self.eq_gain, self.M[0]) * data_recov  # RX:217-218

    # -- a9: one data symbol -----------------------------------------------------------
    def demod_symbol(self, window: np.ndarray) -> np.ndarray:
        # RX:228-230 hands the complex64 stream slice straight to np.fft.fft.  NumPy >= 2.0 then
        # transforms in single precision (NumPy 1.x upcast to complex128); the literal form below
        # follows whichever NumPy is installed, exactly like the reference.  ``force_fp64`` upcasts
        # first: that is the accuracy yardstick the fp32 HIP kernels are measured against.
        w = np.asarray(window)
        if self.force_fp64:
            w = w.astype(np.complex128)
        t_vec = np.fft.fft(w, self.nfft)                                   # RX:230
        x = t_vec[self.bins_used_P]                                        # RX:232
        p_est0 = np.sqrt(len(x) / np.dot(x, np.conj(x)))                   # RX:233
        x = x * p_est0                                                     # RX:235
        x = x * np.exp((1j * (2 * np.pi / self.nfft)) * self.time_synch_ref[1] * self.bins_used_P)  # RX:237-240
        hd = self.est_chan_freq_P[0][self.bins_used_P]                     # RX:242
        g = np.conj(hd) / (1.0 / self.snr_data + hd * np.conj(hd))         # RX:244-246
        return g * x                                                       # RX:248

    def work(self, in0: np.ndarray, out: np.ndarray) -> int:
        """One call of the block (RX:135-262), including its streaming state quirks."""
        in0 = np.asarray(in0)
        n_in = len(in0)
        S, D = self.synch_dat
        N, L = self.nfft, self.rx_b_len
        n_trials = int(np.around(n_in / self.stride_val))                  # RX:139
        n_unique_symb = int(np.floor(n_in / L))                            # RX:140
        n_data_symb = int(n_unique_symb * (D / (S + D)))                   # RX:141
        self.trials_run = 0
        for P in range(n_trials):                                          # RX:143
            if S * L + P * self.stride_val + N + self.start_samp < n_in:   # RX:144
                self.trials_run += 1
                del_mat, y = self.sync_trial(in0, P)
                self.del_mat = del_mat
                dmax_ind = int(np.argmax(np.abs(del_mat)))                 # RX:163
                dmax_val = np.max(np.abs(del_mat))                         # RX:164
                if dmax_val > self.gate * self.MM:                         # RX:166
                    if (P * self.stride_val + self.start_samp - self.time_synch_ref[0] > 2 * self.cp_len + N) \
                            or self.corr_obs == -1:                        # RX:168-169
                        self._accept(P, dmax_ind, dmax_val, y)
                        break                                              # RX:219
        for P in range(0, n_unique_symb, S + D):                           # RX:221
            data_ptr = int(self.time_synch_ref[0] + S * L * (P + 1))       # RX:222
            if self.time_synch_ref[0] + S * L * (P + 1) + N - 1 <= n_in:   # RX:223
                for n_ in range(D):                                        # RX:225
                    start = data_ptr + L * n_
                    self.est_data_freq[P + n_] = self.demod_symbol(in0[start:start + N])   # RX:226-248
        rows = list(range(3, self.est_data_freq.shape[0], S + D))          # RX:249 (literal 3)
        data_demod = np.delete(self.est_data_freq, rows, axis=0)           # RX:249-250
        data_out = np.reshape(data_demod, (1, n_data_symb * self.num_data_bins))   # RX:255
        if self.count > 0:                                                 # RX:257
            out[0:data_out.shape[1]] = data_out[0]                         # RX:258
        self.count += 1                                                    # RX:260
        self.corr_obs = 0                                                  # RX:261
        return len(out)                                                    # RX:262


def rx_work_faithful_ops(rx: RxOracle, in0: np.ndarray) -> None:
    """Same maths as ``RxOracle.work`` (fresh instance) but with the reference's OPERATION
    STRUCTURE: dense ``np.matmul(del_mat_exp, np.diag(y))`` (RX:160), three ``np.diag`` products per
    data symbol (RX:240,244,248), Python list comprehensions.  Used only as the timed
    "reference gr-utsa_ofdm NumPy path" CPU baseline in bench.py."""
    S, D = rx.synch_dat
    N, L = rx.nfft, rx.rx_b_len
    n_in = len(in0)
    n_unique_symb = int(np.floor(n_in / L))
    for P in range(int(np.around(n_in / rx.stride_val))):
        if S * L + P * rx.stride_val + N + rx.start_samp < n_in:
            yv = np.zeros(rx.MM, dtype=complex)
            for LL in range(S):
                a = L * LL + P * rx.stride_val + rx.start_samp
                yv[LL * rx.num_synch_bins:(LL + 1) * rx.num_synch_bins] = \
                    np.fft.fft(in0[a:a + N].astype(np.complex128), N)[rx.synch_bins_used_P]   # RX:148 copies into a c128 buffer
            p_est = np.sqrt(len(yv) / sum(np.multiply(yv, np.conj(yv))))
            yv = p_est * yv
            tmp_2mat = np.matmul(rx.del_mat_exp, np.diag(yv))
            del_mat = np.matmul(tmp_2mat, np.conj(rx.zadoff_chu))
            dmax_ind = int(np.argmax(abs(del_mat)))
            dmax_val = np.max(abs(del_mat))
            if dmax_val > rx.gate * rx.MM:
                rx._accept(P, dmax_ind, dmax_val, yv)
                break
    bp = list(rx.bins_used_P)
    for P in range(0, n_unique_symb, S + D):
        data_ptr = int(rx.time_synch_ref[0] + S * L * (P + 1))
        if rx.time_synch_ref[0] + S * L * (P + 1) + N - 1 <= n_in:
            for n_ in range(D):
                start = data_ptr + L * n_
                t_vec = np.fft.fft(in0[start:start + N], N)
                f0 = t_vec[bp]
                p0 = np.sqrt(len(f0) / (np.dot(f0, np.conj(f0))))
                d0 = f0 * p0
                arg_val = [((1j * (2 * np.pi / N)) * rx.time_synch_ref[1]) * kk for kk in bp]
                dz = np.matmul(np.diag(d0), np.exp(arg_val))
                hd = rx.est_chan_freq_P[0][bp]
                mag = np.matmul(np.diag(hd), np.conj(hd))
                gz = [1.0 / rx.snr_data + vv for vv in mag]
                gq = np.divide(np.conj(hd), gz)
                rx.est_data_freq[P + n_][:] = np.matmul(np.diag(gq), dz)


def rx_demod_frames_vectorised(iq: np.ndarray, frame_len: int, cfg: dict) -> np.ndarray:
    """Honest vectorised CPU baseline for frame-aligned batches (sync hit at P=0 assumed and
    verified): batched FFT over all symbols + elementwise equalise.  Same maths as
    ``RxOracle`` (fresh instance per frame).  Returns est_data_freq of shape
    (n_frames, n_data_sym, Kd) complex128.  Used by bench.py's cpu_baseline leg and tests."""
    N, cp = cfg["nfft"], cfg["cp_len"]
    S, D = cfg["synch_dat"]
    L = N + cp
    n_frames = len(iq) // frame_len
    proto = RxOracle(1, N, cp, cfg["num_synch_bins"], (S, D), cfg["num_data_bins"],
                     cfg["snr"], cfg.get("scale_factor_gate", 0.7))
    n_sym = frame_len // L
    n_pat = n_sym // (S + D)
    fr = np.asarray(iq[:n_frames * frame_len]).reshape(n_frames, frame_len)
    out = np.zeros((n_frames, n_pat * D, cfg["num_data_bins"]), dtype=complex)
    sb, db = proto.synch_bins_used_P, proto.bins_used_P
    zc_c = np.conj(proto.zadoff_chu)
    for f in range(n_frames):
        x = fr[f]
        y = np.concatenate([np.fft.fft(x[L * LL + cp:L * LL + cp + N].astype(np.complex128))[sb]
                            for LL in range(S)])
        y = y * np.sqrt(len(y) / np.sum(y * np.conj(y)))
        c = proto.del_mat_exp @ (y * zc_c)
        d = int(np.argmax(np.abs(c)))
        if not np.max(np.abs(c)) > proto.gate * proto.MM:
            raise RuntimeError("vectorised baseline expects frame-aligned input (sync at P=0)")
        r = proto.del_mat_exp[d] * y
        h = np.sum((r * zc_c / (1 + 1 / proto.snr_ls)).reshape(S, -1), axis=0) / S
        hf = np.zeros(N, dtype=complex)
        hf[sb] = h
        hd = hf[db]
        g = np.conj(hd) / (1 / proto.snr_data + hd * np.conj(hd)) * np.exp(1j * 2 * np.pi / N * d * db)
        # data windows: tsr0 = cp ; ptr = cp + S*L*(P+1) + L*n
        idx = []
        for P in range(0, n_pat * (S + D), S + D):
            for n_ in range(D):
                idx.append(cp + S * L * (P + 1) + L * n_)
        idx = np.asarray(idx)
        win = x[idx[:, None] + np.arange(N)[None, :]].astype(np.complex128)
        X = np.fft.fft(win, axis=1)[:, db]
        X = X * np.sqrt(len(db) / np.sum(X * np.conj(X), axis=1))[:, None]
        out[f] = X * g[None, :]
    return out


# --------------------------------------------------------------------------- CFO-search receiver

# numerology "cases" of G/LEGACY/gr-ofdm-rx/python/SynchEstAndFO.py:36-137
# (num_ofdm_symb, fs, nfft, synch_dat, num_data_bins); cp_len = nfft/4, num_synch_bins = nfft-2, SNR = 1e8 in every case
FO_CASES = {
    0: (48, 960000, 64, (1, 1), 12), 1: (48, 960000, 64, (1, 1), 36), 2: (48, 960000, 64, (1, 1), 48),
    3: (48, 960000, 64, (2, 1), 48), 4: (48, 960000, 64, (3, 1), 24), 5: (48, 960000, 64, (2, 1), 24),
    6: (24, 1920000, 128, (3, 1), 24), 7: (24, 1920000, 128, (5, 1), 100), 8: (12, 3840000, 256, (5, 1), 36),
    9: (12, 3840000, 256, (2, 1), 180),
}


class FoOracle:
    """fp64 restatement of SynchEstAndFO (FO = G/LEGACY/gr-ofdm-rx/python/SynchEstAndFO.py:28-363), Python-2 semantics
    (the file's `/` on ints is floor division: cp_len = nfft/4, corr_size = num_ofdm_symb/sum(synch_dat)).

    Differences from SynchAndChanEst that are kept literally: a table of up to 100 syncs per call and no `break` (FO:250-321);
    one data symbol per sync (FO:324-351); the brute-force carrier-offset search over `fo_range` (FO:261-282); the LS estimate
    uses the sync vector of the LAST candidate while the lag comes from the best one (FO:268-274,300-301); the data symbols are
    rotated with the best candidate of the LAST trial evaluated in the call (`dmax_tmp_ind`, FO:284,331)."""

    MAX_CORR = 100

    CASES = FO_CASES

    def __init__(self, case, fo_range, py2_rotators=True):
        """py2_rotators=True reproduces the only executable semantics of the file: under Python 2, `(1/self.fs)` in FO:192
        is an INTEGER division (fs is an int) = 0, so every carrier-offset rotator is exp(0) = 1 and the search is a no-op
        (all candidates tie, index 0 wins).  py2_rotators=False uses 1.0/fs, the evidently intended rotators (unpinned)."""
        self.num_ofdm_symb, self.fs, self.nfft, sd, self.num_data_bins = self.CASES[case][:5]
        self._setup(sd, self.nfft // 4, self.nfft - 2, 100000000, fo_range, py2_rotators)    # FO:39 (py2 int division)

    @classmethod
    def from_params(cls, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr, force_fp64=False):
        """The same receiver configured by constructor arguments and without a carrier-offset search: this is the work()
        of gr-RXOFDM's synch_and_chan_est (RXc:136-266, "table mode"), which is SynchEstAndFO.work minus the rotators.
        Without the rotator product the data window stays a complex64 slice: np.fft.fft then computes in single precision
        under NumPy >= 2 (the oracle follows the installed NumPy literally; force_fp64=True is the yardstick for the GPU),
        and a short slice is zero-padded by fft(x, N) instead of failing in the product (RXc:228-230)."""
        o = cls.__new__(cls)
        o.num_ofdm_symb, o.fs, o.nfft, o.num_data_bins = int(num_ofdm_symb), 1, int(nfft), int(num_data_bins)
        o._setup(synch_dat, int(cp_len), int(num_synch_bins), snr, [0.0], True)
        o._table_mode = True
        o._force_fp64 = bool(force_fp64)
        return o

    _table_mode = False
    _force_fp64 = False

    def _setup(self, sd, cp_len, num_synch_bins, snr, fo_range, py2_rotators):
        self.synch_dat = [int(sd[0]), int(sd[1])]
        self.cp_len = cp_len
        self.num_synch_bins = num_synch_bins
        self.SNR = snr
        self.fo_range = list(fo_range)
        self.synch_bins_used_P = bins_p(self.num_synch_bins, self.nfft)      # FO:155-158
        self.bins_used_P = bins_p(self.num_data_bins, self.nfft)             # FO:185-187
        self.L_synch = len(self.synch_bins_used_P)
        self.M = [self.synch_dat[0], self.num_synch_bins]
        self.MM = int(np.prod(self.M))
        self.p = 37                                                          # FO:167
        tmp0 = np.arange(self.MM, dtype=np.float64)
        xx = tmp0 * tmp0 if self.num_synch_bins % 2 == 0 else tmp0 * (tmp0 + 1)     # FO:168-173 (parity of Ks)
        self.zadoff_chu = np.exp((-1j * (2 * np.pi / self.MM) * self.p / 2.0) * xx)  # FO:175-176
        inv_fs = (1 // self.fs) if py2_rotators else (1.0 / self.fs)
        self.cfo = np.exp(1j * 2 * np.pi * inv_fs * np.outer(self.fo_range, np.arange(self.nfft)))         # FO:192
        self.del_mat_exp = np.tile(np.exp((1j * (2.0 * np.pi / self.nfft)) * np.outer(
            np.arange(self.cp_len + 1), self.synch_bins_used_P)), (1, self.M[0]))    # FO:193-194
        self.stride_val = self.cp_len - 1                                    # FO:196
        self.start_samp = self.cp_len
        self.rx_b_len = self.nfft + self.cp_len
        n = self.MAX_CORR
        self.time_synch_ref = np.zeros((n, 3))                               # FO:202
        self.est_chan_time = np.zeros((n, self.nfft), dtype=complex)
        self.est_synch_freq = np.zeros((n, self.MM), dtype=complex)
        self.est_chan_freq_P = np.zeros((n, self.nfft), dtype=complex)
        self.est_data_freq = np.zeros((n, self.num_data_bins), dtype=complex)
        self.cor_obs = -1
        self.count = 0
        self.dmax_tmp_ind = None
        self.eq_gain = None

    def work(self, in0, out):
        in0 = np.asarray(in0)
        self._loop_a(in0)
        self._loop_b(in0)
        corr_size = self.num_ofdm_symb // sum(self.synch_dat)                # FO:362 (py2 int division)
        data_out = np.reshape(self.est_data_freq[0:corr_size], (1, corr_size * self.num_data_bins))   # FO:364
        if self.count > 0:
            out[0:data_out.shape[1]] = data_out[0]                           # FO:366-367
        self.count += 1
        self.cor_obs = 0                                                     # FO:369
        return len(out)

    def _loop_a(self, in0):
        n_in = len(in0)
        S, N, L, cp = self.M[0], self.nfft, self.rx_b_len, self.cp_len
        zc_c = np.conj(self.zadoff_chu)
        n_trials = int(np.around(n_in / self.stride_val))                    # FO:246
        for P in range(n_trials):                                            # FO:248
            if S * L + P * self.stride_val + N + self.start_samp < n_in:     # FO:249
                win = [in0[L * LL + P * self.stride_val + cp: L * LL + P * self.stride_val + cp + N].astype(np.complex128)
                       for LL in range(S)]                                   # FO:250-253
                dmax_ind0 = np.zeros(len(self.fo_range), dtype=int)
                dmax_val0 = np.zeros(len(self.fo_range))
                y = None
                for fo in range(len(self.fo_range)):                         # FO:258
                    y = np.concatenate([np.fft.fft(win[LL] * self.cfo[fo], N)[self.synch_bins_used_P] for LL in range(S)])
                    y = y * np.sqrt(len(y) / np.sum(y * np.conj(y)))         # FO:268-270
                    del_mat = self.del_mat_exp @ (y * zc_c)                  # FO:272-275
                    dmax_ind0[fo] = np.argmax(np.abs(del_mat))               # FO:277
                    dmax_val0[fo] = np.max(np.abs(del_mat))                  # FO:278
                dmax_val = np.max(dmax_val0)                                 # FO:282
                self.dmax_tmp_ind = int(np.argmax(dmax_val0))                # FO:283
                dmax_ind = dmax_ind0[self.dmax_tmp_ind]                      # FO:285
                if dmax_val > 0.4 * self.MM:                                 # FO:288
                    tim_synch_ind = self.time_synch_ref[max(self.cor_obs, 0)][0]     # FO:289
                    if (P * self.stride_val + self.start_samp - tim_synch_ind > 2 * cp + N) or self.cor_obs == -1:
                        self.cor_obs += 1                                    # FO:294 (IndexError past 100 rows)
                        self.time_synch_ref[self.cor_obs] = [P * self.stride_val + self.start_samp, dmax_ind, int(dmax_val)]
                        data_recov = self.del_mat_exp[dmax_ind] * y          # FO:300-301: y of the LAST candidate
                        tmp_v1 = data_recov * zc_c / (1.0 / self.SNR + 1.0)  # FO:303-304
                        chan_est = np.sum(tmp_v1.reshape(S, self.L_synch), axis=0) / float(S)    # FO:306-307
                        chan_est1 = np.zeros(N, dtype=complex)
                        chan_est1[self.synch_bins_used_P] = chan_est         # FO:309-311
                        self.est_chan_freq_P[self.cor_obs] = chan_est1
                        self.est_chan_time[self.cor_obs] = np.fft.ifft(chan_est1, N)          # FO:313,323
                        self.eq_gain = np.conj(chan_est) / (1.0 / self.SNR + chan_est * np.conj(chan_est))   # FO:324-327
                        self.est_synch_freq[self.cor_obs] = np.tile(self.eq_gain, S) * data_recov           # FO:328-329
    def _demod_row(self, in0, P):
        """One equalised data symbol of sync P (FO:335-358); None when the guard FO:334 fails."""
        S, N, L = self.M[0], self.nfft, self.rx_b_len
        if not (self.time_synch_ref[P][0] + S * L + N - 1 <= len(in0)):      # FO:334
            return None
        data_ptr = int(self.time_synch_ref[P][0] + S * L)                    # FO:335
        if self._table_mode:
            x = in0[data_ptr: data_ptr + N]                                  # RXc:228
            if self._force_fp64:
                x = x.astype(np.complex128)
        else:
            x = in0[data_ptr: data_ptr + N] * self.cfo[self.dmax_tmp_ind]    # FO:338-339 (ValueError if the slice is short)
        t_vec = np.fft.fft(x, N)                                             # FO:340 / RXc:230
        f0 = t_vec[self.bins_used_P]
        f0 = f0 * np.sqrt(len(f0) / np.dot(f0, np.conj(f0)))                 # FO:343-345
        f0 = f0 * np.exp((1j * (2 * np.pi / N)) * self.time_synch_ref[P][1] * self.bins_used_P)   # FO:347-350
        hd = self.est_chan_freq_P[P][self.bins_used_P]                       # FO:352
        return np.conj(hd) / (1.0 / self.SNR + hd * np.conj(hd)) * f0        # FO:354-358

    def _loop_b(self, in0):
        for P in range(self.cor_obs + 1):                                    # FO:332
            row = self._demod_row(in0, P)
            if row is not None:
                self.est_data_freq[P] = row


# numerology of G/LEGACY/gr-ofdm-rx/python/SynchEstFOAndDSSS.py:37-157 (DS = that file)
# case -> (num_ofdm_symb, fs, nfft, synch_dat, num_data_bins, DSSS); cp_len = nfft/4, num_synch_bins = nfft-2, SNR = 1e8
DSSS_CASES = {
    0: (48, 960000, 64, (2, 1), 12, 1), 1: (48, 960000, 64, (3, 1), 36, 6), 2: (45, 960000, 64, (4, 1), 48, 6),
    3: (45, 960000, 64, (4, 1), 48, 12), 4: (24, 1920000, 128, (3, 1), 32, 8), 5: (20, 1920000, 128, (4, 1), 84, 12),
    6: (20, 1920000, 128, (4, 1), 96, 16), 7: (24, 1920000, 128, (5, 1), 120, 24), 8: (12, 3840000, 256, (3, 1), 168, 12),
    9: (10, 3840000, 256, (4, 1), 192, 16), 10: (10, 3840000, 256, (4, 1), 240, 24),
}


def spreading_code(dsss: int, p: int = 37) -> np.ndarray:
    """ZC-form spreading sequence of length DSSS (DS:253-262)."""
    t0 = np.arange(dsss, dtype=np.float64)
    xx = t0 * t0 if dsss % 2 == 0 else t0 * (t0 + 1)
    return np.exp((-1j * (2 * np.pi / dsss) * p / 2.0) * xx)


class FoDsssOracle(FoOracle):
    """SynchEstFOAndDSSS (DS:28-413) = SynchEstAndFO with its own case table plus: every equalised data symbol is despread
    over groups of DSSS consecutive listed bins with conj(SC) and averaged (DS:391-399); the output is the despread rows and
    is written on EVERY call (the count gate is commented out, DS:405-407).  Kept literally: the statement DS:392 re-assigns
    est_data_freq[P] from the loop's locals outside the guard, so a row whose guard fails before any row passed raises
    UnboundLocalError (only row 0 of an earlier call can do that)."""

    CASES = DSSS_CASES

    def __init__(self, case, fo_range, py2_rotators=True):
        FoOracle.__init__(self, case, fo_range, py2_rotators)
        self.DSSS = self.CASES[case][5]
        self.SC = spreading_code(self.DSSS, self.p)
        self.est_data_freq_d = np.zeros((self.MAX_CORR, int(self.num_data_bins // self.DSSS)), dtype=complex)   # DS:243

    def _loop_b(self, in0):
        n_spread = int(len(self.bins_used_P) // self.DSSS)                   # DS:358
        last = None
        for P in range(self.cor_obs + 1):                                    # DS:360
            row = self._demod_row(in0, P)
            if row is not None:
                last = row
            if last is None:
                raise UnboundLocalError("local variable 'data_recov_z' referenced before assignment")   # DS:392
            self.est_data_freq[P] = last                                     # DS:392
            grp = self.est_data_freq[P][:n_spread * self.DSSS].reshape(n_spread, self.DSSS)
            self.est_data_freq_d[P] = np.mean(grp * np.conj(self.SC)[None, :], axis=1)          # DS:393-397

    def work(self, in0, out):
        in0 = np.asarray(in0)
        self._loop_a(in0)
        self._loop_b(in0)
        corr_size = self.num_ofdm_symb // sum(self.synch_dat)                # DS:401
        data_out = np.reshape(self.est_data_freq_d[0:corr_size], (1, corr_size * self.est_data_freq_d.shape[1]))
        out[0:data_out.shape[1]] = data_out[0]                               # DS:407 (no count gate)
        self.count += 1
        self.cor_obs = 0
        return len(out)


# --------------------------------------------------------------------------- regression-tracking receiver
# SE = G/LEGACY/gr-ofdm-rx/python/SynchronizeAndEstimate.py
TRACKER_PROFILES = {
    # case -> (channel_band, bin_spacing, num_ant_txrx, SNR, num_symbols[0])   (SE:33-60)
    0: (0.97 * 960e3, 15e3, 1, 100, 48),
    1: (0.9 * 20e6, 312.5e3, 2, 50, 10),
}


class TrackerOracle:
    """fp64 restatement of SynchronizeAndEstimate (SE:25-442): acquisition by a strided ZC lag-correlation search, then one
    sync symbol per [1,3] pattern at a pointer that is first advanced by the pattern length and, from the sixth sync on,
    predicted by a least-squares line through the last five (position + lag) observations (SE:333-350); LS channel estimate
    per sync; three equalised data symbols per sync, each renormalised by the power of est_data_freq row `p` (SE:431-434).

    Kept literally: the lag is reported minus one (`dmax_ind = dmax_ind0 - 1`, SE:275) and a lag of -1 indexes the LAST column
    of the phase matrix in the estimate (SE:354) but de-rotates the data by -1 (SE:421); the "late lag" branch moves the
    pointer by cp/2 but recomputes the FFT of the SAME window (SE:281-309); the distance rule reads row max(corr_obs, 1),
    which is a row of the PREVIOUS call for the first two syncs (SE:311-313); time_synch_ref[.., 2] holds the raw peak.
    Case 1 (two antennas) runs the sync stage only (SE:397: the data stage is single-antenna)."""

    def __init__(self, case):
        band, spacing, self.num_ant_txrx, self.SNR, n_symb = TRACKER_PROFILES[case]
        self.synch_data = np.array([1, 3])
        self.NFFT = int(2 ** (np.ceil(np.log2(round(band / spacing)))))          # SE:83
        self.len_CP = int(round(self.NFFT / 4))                                  # SE:85
        num_bins1 = 4 * np.floor(np.floor(band / spacing) / 4)                   # SE:87-90
        all_bins = np.array(list(range(-int(num_bins1 / 2), 0)) + list(range(1, int(num_bins1 / 2) + 1)))
        self.num_data_bins = len(all_bins)                                       # ref_sigs = 0: no pilot bins (SE:80,95-101)
        self.used_bins_data = ((self.NFFT + all_bins) % self.NFFT).astype(int)   # SE:102
        n_pat = int(np.ceil(n_symb / sum(self.synch_data)))                      # SE:104
        self.lmax_s, self.lmax_d = n_pat * 1, n_pat * 3                          # SE:143-144
        self.rx_buff_len = self.NFFT + self.len_CP
        self.num_synch_bins = self.NFFT - 2
        self.MM = int(self.synch_data[0] * self.num_synch_bins)
        self.synch_ref = zadoff_chu(self.MM, 23)                                 # SE:123-130
        self.used_bins_synch = bins_p(self.num_synch_bins, self.NFFT)            # SE:134-136
        a, N = self.num_ant_txrx, self.NFFT
        self.est_chan_freq_p = np.zeros((a, self.lmax_s, N), dtype=complex)
        self.est_chan_freq_n = np.zeros((a, self.lmax_s, self.num_synch_bins), dtype=complex)
        self.est_synch_freq = np.zeros((a, self.lmax_s, self.num_synch_bins), dtype=complex)
        self.est_chan_impulse = np.zeros((a, self.lmax_s, N), dtype=complex)
        if a == 1:
            self.est_data_freq = np.zeros((a, self.lmax_d, self.num_data_bins), dtype=complex)
        self.time_synch_ref = np.zeros((a, 250, 3))                              # SE:179
        self.corr_obs = None
        self.force_fp64 = False

    def _trial(self, in0, ptr_frame):
        """Window at ptr_frame -> (normalised sync-bin vector, phase matrix, peak, arg-peak)  (SE:236-276)."""
        N = self.NFFT
        start = int(ptr_frame)
        w = np.zeros(N, dtype=complex)
        w[:] = in0[start:start + N]                                              # SE:240-243 (cast to complex128)
        f = np.fft.fft(w, N)[self.used_bins_synch]
        pow_est = np.sum(f * np.conj(f)).real / len(f)                           # SE:255
        synch_dat = f / np.sqrt(pow_est)
        p_mat = np.exp(1j * 2 * (np.pi / N) * np.outer(self.used_bins_synch, np.arange(self.len_CP + 1)))   # SE:264-268
        del_mat = np.conj(self.synch_ref) @ (synch_dat[:, None] * p_mat)         # SE:271
        dd = np.abs(del_mat)
        return synch_dat, p_mat, dd.max(), int(dd.argmax())

    def work(self, in0, out):
        in0 = np.asarray(in0)
        n_in = in0.shape[0]
        N, cp, L, m = self.NFFT, self.len_CP, self.rx_buff_len, 0
        stride_val = np.ceil(cp / 2)                                             # SE:209
        ptr_frame, b, xp = 0, 0, []
        self.corr_obs = -1                                                       # SE:216
        start_samp = (cp - 4) - 1                                                # SE:219
        total_loops = int(np.ceil(n_in / stride_val))
        ptr_adj, loop_count, sym_count = 0, 0, 0
        tap_delay = 5
        x = np.zeros(tap_delay)
        ptr_synch0 = np.zeros(1000)
        sd = int(sum(self.synch_data))
        while loop_count <= total_loops:                                         # SE:230
            if self.corr_obs == -1:
                ptr_frame = loop_count * stride_val + start_samp + ptr_adj
            elif self.corr_obs < 5:
                ptr_frame += sd * (N + cp)
            else:
                ptr_frame = (np.ceil(np.dot(xp[-1:], b) - cp / 4))[0]           # SE:237
            if N + ptr_frame < n_in:                                             # SE:240 (M[0] = 1)
                synch_dat, p_mat, dmax, dmax_ind0 = self._trial(in0, ptr_frame)
                dmax_ind = dmax_ind0 - 1                                         # SE:275
                if dmax > 0.5 * len(synch_dat) or self.corr_obs > -1:            # SE:279
                    if dmax_ind > np.ceil(0.75 * cp):                            # SE:281: pointer moves, window does not
                        if self.corr_obs == -1:
                            ptr_adj += np.ceil(0.5 * cp)
                            ptr_frame = loop_count * stride_val + start_samp + ptr_adj
                        elif self.corr_obs < 5:
                            ptr_frame += np.ceil(0.5 * cp)
                    time_synch_ind = self.time_synch_ref[m, max(self.corr_obs, 1), 0]           # SE:311
                    if ptr_frame - time_synch_ind > (2 * cp + N) or self.corr_obs == -1:        # SE:313
                        self.corr_obs += 1
                        self.time_synch_ref[m, self.corr_obs] = [ptr_frame, dmax_ind, dmax]     # SE:316-318
                        ptr_synch0[sym_count % tap_delay] = sum(self.time_synch_ref[m, self.corr_obs, 0:2])
                        x[sym_count % tap_delay] = sym_count * sd
                        sym_count += 1
                        x2 = x[0:min(self.corr_obs, tap_delay)]
                        x_plus = np.concatenate((x2, np.atleast_1d(sym_count * sd)))
                        xp = np.zeros((len(x_plus), 2))
                        xp[:, 0] = 1
                        xp[:, 1] = x_plus
                        if self.corr_obs > 3:                                    # SE:333-341
                            y = ptr_synch0[0:min(tap_delay, self.corr_obs)]
                            X = np.zeros((len(x2), 2))
                            X[:, 0] = 1
                            X[:, 1] = x2
                            b = np.linalg.lstsq(X, y, rcond=-1)[0]               # the reference's (legacy-default) rcond
                        data_recov0 = synch_dat * p_mat[:, dmax_ind]             # SE:344 (-1 -> last column)
                        h_est = (data_recov0 * np.conj(self.synch_ref)) / (1 + (1 / self.SNR))   # SE:348-353 (M[0] = 1)
                        h_est1 = np.zeros(N, dtype=complex)
                        h_est1[self.used_bins_synch] = h_est
                        self.est_chan_freq_p[m, self.corr_obs] = h_est1          # IndexError past lmax_s rows
                        self.est_chan_freq_n[m, self.corr_obs] = h_est
                        self.est_chan_impulse[m, self.corr_obs] = np.fft.ifft(h_est1, N)        # SE:369-370
                        self.est_synch_freq[m, self.corr_obs] = (data_recov0 * np.conj(h_est)) / (
                            (np.conj(h_est) * h_est) + (1 / self.SNR))           # SE:375-378
            loop_count += 1
        if self.num_ant_txrx == 1:                                               # SE:397
            D = int(self.synch_data[1])
            for p in range(self.corr_obs + 1):
                for data_sym in range(D):
                    if sum(self.time_synch_ref[m, p, :]) + N < n_in:             # SE:401 (pointer + lag + PEAK)
                        data_ptr = int(self.time_synch_ref[m, p, 0] + (data_sym + 1) * L)
                        seg = in0[data_ptr: data_ptr + N]                        # complex64 slice; fft(x, N) zero-pads
                        if self.force_fp64:
                            seg = seg.astype(np.complex128)
                        freq_dat0 = np.fft.fft(seg, N)[self.used_bins_data]
                        # builtin sum, as the reference: on a complex64 vector (NumPy >= 2) it also ACCUMULATES in single
                        # precision; force_fp64 is the yardstick for the GPU
                        p_est = sum(freq_dat0 * np.conj(freq_dat0)) / len(freq_dat0)             # SE:411
                        data_recov0 = freq_dat0 / np.sqrt(p_est)
                        h_est = self.est_chan_freq_p[m, p, self.used_bins_data]
                        data_recov = data_recov0 * np.exp(
                            1j * 2 * (np.pi / N) * self.used_bins_data * self.time_synch_ref[m, p, 1])   # SE:420-422
                        row = p * D + data_sym
                        self.est_data_freq[m, row] = (data_recov * np.conj(h_est)) / ((np.conj(h_est) * h_est) + (1 / self.SNR))
                        data = self.est_data_freq[m, p]                          # SE:431: row p, not the row just written
                        p_est1 = sum(data * np.conj(data)) / len(data)
                        self.est_data_freq[m, row] = self.est_data_freq[m, row] / np.sqrt(p_est1)
                        out[0:self.num_data_bins] = self.est_data_freq[m, row]   # SE:438-440: every symbol lands at out[0:Kd]
        return len(out)
