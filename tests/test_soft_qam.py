"""16/64-QAM soft de-map: the extension of BitRecovery's metric defined in oracle.soft_demap_qam.
PARITY UNPINNED against the reference (it implements QPSK only): the oracle is checked through properties here, the HIP
kernels against the oracle (fp32 metrics within 1e-5 norm-relative, hard bits exact)."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ofdm_oracle as orc

TOL = 1e-5


def _noisy(mod, n, sigma, seed):
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, n * orc.BITS_PER_SYMBOL[mod])
    x = orc.map_bits(bits, mod)
    z = x + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return bits, z.astype(np.complex64)


@pytest.mark.parametrize("mod", ["16QAM", "64QAM"])
def test_soft_qam_oracle_properties(mod):
    bits, z = _noisy(mod, 4000, 0.02, 5)
    hard, s0, s1 = orc.soft_demap_qam(z, mod)
    assert np.array_equal(hard, bits)                                      # clean enough: no errors
    assert np.array_equal(hard, orc.demap_hard(z, mod))                    # sign rule == decision regions
    assert (s0 <= 0).all() and (s1 <= 0).all()
    # the decided hypothesis carries the distance to the nearest level: identical for every bit of an axis
    bps = orc.BITS_PER_SYMBOL[mod]
    best = np.maximum(s0, s1).reshape(-1, bps)
    assert np.allclose(best[:, 0::2], best[:, 0:1]) and np.allclose(best[:, 1::2], best[:, 1:2])
    # sigma: 0.7071 * mean distance to the nearest point
    lv, _ = orc.qam_levels(mod)
    zz = z.astype(np.complex128)
    dmin = np.hypot(np.min(np.abs(zz.real[:, None] - lv), axis=1), np.min(np.abs(zz.imag[:, None] - lv), axis=1))
    hf = -0.5 / (0.7071067811865476 * dmin.mean()) ** 2
    assert np.allclose(best[:, 0], hf * np.min(np.abs(zz.real[:, None] - lv), axis=1))
    # scale covariance: symbols on a noisier buffer get metrics that shrink with 1/sigma^2
    _, z2 = _noisy(mod, 4000, 0.04, 5)
    _, t0, _ = orc.soft_demap_qam(z2, mod)
    assert 0.15 < np.mean(np.abs(t0)) / np.mean(np.abs(s0)) < 0.4


def test_soft_qam_reduces_to_reference_rule_for_qpsk_inliers():
    """The per-axis rule (distance to the nearest level with the bit value) is BitRecovery's |e| / K-|e| pair whenever the
    coordinate lies between the two QPSK levels."""
    _, z = _noisy("QPSK", 3000, 0.05, 9)
    _, s0, s1 = orc.bit_recovery(z)
    zz = z.astype(np.complex128)
    lv = np.array([0.7071067811865476, -0.7071067811865476])               # bit 0 -> +, bit 1 -> -
    x = np.stack([zz.real, zz.imag], axis=1).ravel()
    inl = np.abs(x) <= lv[0]
    e = np.min(np.abs(x[:, None] - lv), axis=1)
    dmin = np.hypot(e[0::2], e[1::2])
    hf = -0.5 / (0.7071067811865476 * dmin.mean()) ** 2
    assert np.allclose(s0[inl], hf * np.abs(x - lv[0])[inl], rtol=1e-9, atol=1e-9)
    assert np.allclose(s1[inl], hf * np.abs(x - lv[1])[inl], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("mod,sigma", [("16QAM", 0.05), ("64QAM", 0.03), ("16QAM", 0.5), ("64QAM", 0.4)])
def test_soft_qam_gpu_vs_oracle(mod, sigma):
    import ofdm_mi355x as om
    bps = orc.BITS_PER_SYMBOL[mod]
    _, z = _noisy(mod, 70001, sigma, 3)
    hard, s0, s1 = orc.soft_demap_qam(z, mod)
    rx = om.RxEngine(1, 64, 16, 62, (1, 3), 60, 100)
    d_z = om.DeviceBuffer(z.nbytes).upload(z)
    d_h = om.DeviceBuffer(len(z) * bps)
    d_0 = om.DeviceBuffer(len(z) * bps * 4)
    d_1 = om.DeviceBuffer(len(z) * bps * 4)
    rx.demap(d_z, len(z), mod, d_h, d_0, d_1)
    g0 = d_0.download(np.float32, len(z) * bps)
    g1 = d_1.download(np.float32, len(z) * bps)
    gh = d_h.download(np.uint8, len(z) * bps)
    assert relerr(g0, s0) < TOL and relerr(g1, s1) < TOL
    # hard bits: exact wherever the two metrics are not tied to fp32 rounding (a coordinate on a decision edge)
    clear = np.abs(s1 - s0) > 1e-4 * np.abs(s1 + s0)
    assert clear.mean() > 0.999
    assert np.array_equal(gh[clear], hard[clear])
    assert np.array_equal(gh, orc.demap_hard(z, mod))


@pytest.mark.gpu
@pytest.mark.parametrize("mod", ["16QAM", "64QAM"])
def test_bitrecovery_block_qam(mod):
    import OFDMReceiver
    bits, z = _noisy(mod, 6000, 0.03, 8)
    blk = OFDMReceiver.BitRecovery(mod, "/tmp/", 0)
    assert blk.work([z], [None]) == len(z)
    hard, s0, s1 = orc.soft_demap_qam(z, mod)
    assert np.array_equal(blk.hardbit.ravel(), bits)
    assert relerr(blk.softbit0, s0) < TOL and relerr(blk.softbit1, s1) < TOL
