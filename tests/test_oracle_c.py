"""Pins the plain-C scalar restatement (oracle/ofdm_oracle_c.c) to the recorded reference runs and to the NumPy oracle.  CPU only.

The recorded `est_data_freq` rows come from the reference under NumPy >= 2, whose `np.fft.fft` of the complex64 slice
(`SynchAndChanEst.py:230`) runs in single precision: fp32-level agreement there, fp64-level agreement with
`RxOracle(force_fp64=True)` and with the recorded sync-stage arrays (which the reference computes in complex128)."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ofdm_oracle as orc
from oracle import oracle_c

SYNTH = ["n64_lead5", "n256", "n1024_lead3", "n2048", "n2048_snr30", "n4096"]


@pytest.mark.parametrize("tag", SYNTH)
def test_c_restatement_matches_reference_run_synthetic(golden, tag):
    g = golden("ref_rx_synth.npz")
    N, cp, Kd, n_sym, lead, snr = (int(v) for v in g[tag + "_cfg"])
    iq = g[tag + "_iq"]
    gate = float(g[tag + "_gate"][0])
    tsr, H, edf, trials = oracle_c.rx_work(iq, n_sym, N, cp, N - 2, (1, 3), Kd, snr, gate)
    assert np.array_equal(tsr, g[tag + "_tsr"])
    assert relerr(H, g[tag + "_H"]) < 1e-11
    assert relerr(edf, g[tag + "_edf"]) < 2e-6
    rx = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, snr, gate, force_fp64=True)
    rx.work(iq, np.zeros(len(iq), np.complex64))
    assert relerr(edf, rx.est_data_freq) < 1e-10 and relerr(H, rx.est_chan_freq_P[0]) < 1e-11
    assert trials == rx.trials_run


@pytest.mark.parametrize("tag", ["offline", "online"])
def test_c_restatement_matches_reference_run_on_fixture(golden, tag):
    fx = golden("ref_fixtures.npz")
    ref = golden("ref_rx_fixture64.npz")
    iq = fx["tx_" + tag][0].astype(np.complex64)
    tsr, H, edf, _ = oracle_c.rx_work(iq, 240, 64, 16, 62, (1, 3), 60, 100, 0.7)
    assert np.array_equal(tsr, ref[tag + "_tsr"])
    assert relerr(H, ref[tag + "_H"]) < 1e-11
    assert relerr(edf, ref[tag + "_edf"]) < 2e-6
    rows = [r for r in range(240) if r % 4 != 3]
    assert np.array_equal(orc.demap_hard(edf[rows].ravel(), "QPSK"), fx["tx_bits"][0].astype(np.uint8))      # 0 / 21600


def test_c_restatement_guard_zero_rows_and_padding():
    """Late sync: the last pattern fails the guard (zero rows), a window past the end is zero-padded -- as the NumPy oracle."""
    N, cp, Kd, n_sym = 64, 16, 60, 12
    L = N + cp
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, 9 * Kd * 2).astype(np.uint8)
    tx = orc.channel_apply(orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym), orc.REF_TAPS, N)
    for lead in (0, L // 2, 2 * L + cp + 16):
        fl = n_sym * L + 7
        x = np.concatenate([0.3 * (rng.standard_normal(lead) + 1j * rng.standard_normal(lead)), tx])[:fl]
        x = (np.concatenate([x, np.zeros(fl - len(x))]) + 0.02 * (rng.standard_normal(fl) + 1j * rng.standard_normal(fl))).astype(np.complex64)
        rx = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, 30, 0.7, force_fp64=True)
        rx.work(x, np.zeros(fl, np.complex64))
        tsr, H, edf, trials = oracle_c.rx_work(x, n_sym, N, cp, N - 2, (1, 3), Kd, 30, 0.7)
        assert np.array_equal(tsr, rx.time_synch_ref) and trials == rx.trials_run
        assert np.array_equal(edf.any(axis=1), rx.est_data_freq.any(axis=1))
        assert relerr(edf, rx.est_data_freq) < 1e-10
    with pytest.raises(IndexError):
        oracle_c.rx_work(x, 4, N, cp, N - 2, (1, 3), Kd, 30, 0.7)


def test_c_oracle_frames_entry_equals_the_single_frame_entry():
    """The OpenMP multi-frame entry (CPU baseline, every usable core) returns the single-frame entry's sync reference per frame."""
    from oracle import oracle_c
    N, cp, Kd, n_sym = 64, 16, 60, 8
    rng = np.random.default_rng(3)
    frames = []
    for lead in (0, 5, 33):
        tx = orc.channel_apply(orc.tx_modulate(rng.integers(0, 2, 6 * Kd * 2), N, cp, N - 2, Kd, n_sym), orc.REF_TAPS, N)
        x = np.concatenate([0.01 * (rng.standard_normal(lead) + 1j * rng.standard_normal(lead)), tx])[:n_sym * (N + cp)]
        frames.append(np.concatenate([x, np.zeros(n_sym * (N + cp) - len(x))]).astype(np.complex64))
    iq = np.stack(frames)
    tsr, hits = oracle_c.rx_work_frames(iq, n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7, n_threads=2)
    assert hits == 3
    for f in range(3):
        t1, _, _, _ = oracle_c.rx_work(iq[f], n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7)
        assert np.array_equal(tsr[f], t1)

