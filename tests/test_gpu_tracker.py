"""GPU parity of the regression-tracking receiver: OFDMReceiver.SynchronizeAndEstimate (host pointer logic +
device primitives through the C ABI) vs recorded runs of the reference block (tests/golden/ref_tracker.npz) and vs
oracle.TrackerOracle on seeded inputs.  Pointers and lags exact, peaks and fp32 array outputs within 1e-5 norm-relative."""
import warnings

import numpy as np
import pytest

from conftest import relerr
from oracle import ofdm_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _block(case):
    import OFDMReceiver
    return OFDMReceiver.SynchronizeAndEstimate(case)


def _compare(blk, tsr, Hp, Hn, imp, esf, edf, out, ref_out):
    assert np.array_equal(blk.time_synch_ref[:, :, 0:2], tsr[:, :, 0:2])
    assert relerr(blk.time_synch_ref[:, :, 2], tsr[:, :, 2]) < TOL
    assert relerr(blk.est_chan_freq_p, Hp) < TOL
    assert relerr(blk.est_chan_freq_n, Hn) < TOL
    assert relerr(blk.est_chan_impulse, imp) < TOL
    assert relerr(blk.est_synch_freq, esf) < TOL
    if edf is not None:
        assert relerr(blk.est_data_freq, edf) < TOL
        assert relerr(out, ref_out) < TOL
    else:
        assert not out.any() and not ref_out.any()


@pytest.mark.parametrize("tag", ["clean", "fade", "noisy", "mimo_cfg", "txdata0"])
def test_tracker_block_on_reference_runs(golden, tag):
    g = golden("ref_tracker.npz")
    case = int(g[tag + "_case"][0])
    blk = _block(case)
    iq = g[tag + "_iq"]
    warnings.simplefilter("ignore")
    for call in (1, 2):
        out = np.zeros(len(iq), np.complex64)
        assert blk.work([iq], [out]) == len(iq)
        k = "%s_call%d_" % (tag, call)
        assert blk.corr_obs == int(g[k + "corr_obs"][0])
        _compare(blk, g[k + "tsr"], g[k + "Hp"], g[k + "Hn"], g[k + "imp"], g[k + "esf"],
                 g[k + "edf"] if case == 0 else None, out, g[k + "out"])
    if case == 0 and tag != "txdata0":
        blk2 = _block(0)
        blk2.work([iq], [np.zeros(len(iq), np.complex64)])
        n_sync = blk2.corr_obs + 1
        got = orc.demap_hard(blk2.est_data_freq[0][:n_sync * 3].ravel(), "QPSK")
        assert np.array_equal(got, g[tag + "_bits"][:n_sync * 3 * 120])


@pytest.mark.parametrize("seed,lead,sigma,n_sym", [(1, 0, 0.0, 48), (2, 13, 0.05, 40), (3, 29, 0.01, 48), (4, 5, 0.0, 20)])
def test_tracker_block_vs_oracle(seed, lead, sigma, n_sym):
    rng = np.random.default_rng(seed)
    n_data = sum(1 for s in range(n_sym) if s % 4 >= 1)
    bits = rng.integers(0, 2, n_data * 120)
    tx = orc.channel_apply(orc.tx_modulate(bits, 64, 16, 62, 60, n_sym, synch_dat=(1, 3), zc_root=23), orc.REF_TAPS, 64)
    tx = tx[:n_sym * 80 + 8] + sigma * (rng.standard_normal(n_sym * 80 + 8) + 1j * rng.standard_normal(n_sym * 80 + 8))
    iq = np.concatenate([np.zeros(lead), tx, np.zeros(50)]).astype(np.complex64)
    o = orc.TrackerOracle(0)
    o.force_fp64 = True
    blk = _block(0)
    warnings.simplefilter("ignore")
    for _ in range(3):
        ro, rb = np.zeros(len(iq), np.complex64), np.zeros(len(iq), np.complex64)
        o.work(iq, ro)
        blk.work([iq], [rb])
        assert blk.corr_obs == o.corr_obs
        _compare(blk, o.time_synch_ref, o.est_chan_freq_p, o.est_chan_freq_n, o.est_chan_impulse, o.est_synch_freq,
                 o.est_data_freq, rb, ro)


def test_tracker_block_index_error_past_the_pattern_rows():
    """A buffer with more sync symbols than est_chan_freq_p has rows: IndexError at the 13th sync, like the reference."""
    rng = np.random.default_rng(9)
    bits = rng.integers(0, 2, 45 * 120)
    iq = np.concatenate([orc.tx_modulate(bits, 64, 16, 62, 60, 60, synch_dat=(1, 3), zc_root=23), np.zeros(40)]).astype(np.complex64)
    o = orc.TrackerOracle(0)
    warnings.simplefilter("ignore")
    with pytest.raises(IndexError):
        o.work(iq, np.zeros(len(iq), np.complex64))
    blk = _block(0)
    with pytest.raises(IndexError):
        blk.work([iq], [np.zeros(len(iq), np.complex64)])
    assert blk.corr_obs == o.corr_obs == 12
    assert np.array_equal(blk.time_synch_ref[:, :, 0:2], o.time_synch_ref[:, :, 0:2])
