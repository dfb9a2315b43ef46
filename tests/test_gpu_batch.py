"""GPU parity of the frame-batched device path, TX chain, channel and de-mapper (run with -m gpu)."""
import numpy as np
import pytest

from conftest import assert_close, poisoned, relerr
from oracle import ofdm_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def om():
    import ofdm_mi355x
    ofdm_mi355x.load()
    return ofdm_mi355x


def _frames(N, cp, Kd, n_sym, n_frames, mod, seed, tail):
    """oracle TX + reference 5-tap channel per frame -> (bits[n_frames, nb], iq complex64 [n_frames, frame_len])"""
    rng = np.random.default_rng(seed)
    bps = orc.BITS_PER_SYMBOL[mod]
    nb = (n_sym // 4) * 3 * Kd * bps
    L = N + cp
    bits = rng.integers(0, 2, (n_frames, nb)).astype(np.uint8)
    iq = np.zeros((n_frames, n_sym * L + tail), np.complex64)
    for f in range(n_frames):
        tx = orc.tx_modulate(bits[f], N, cp, N - 2, Kd, n_sym, modulation=mod)
        iq[f] = orc.channel_apply(tx, orc.REF_TAPS, N)[:n_sym * L + tail]
    return bits, iq


@pytest.mark.parametrize("N,cp,Kd,mod,n_frames,n_sym", [
    (64, 16, 60, "QPSK", 19, 12),          # odd frame count: partially filled workgroups (8 symbols per wave)
    (128, 32, 100, "QPSK", 7, 8),
    (256, 18, 152, "16QAM", 5, 8),
    (512, 36, 300, "64QAM", 3, 8),
    # BASELINE.json configs[4]: every cell of {1024,2048,4096}-pt x {QPSK,16-QAM,64-QAM}
    (1024, 72, 600, "QPSK", 3, 8), (1024, 72, 600, "16QAM", 3, 12), (1024, 72, 600, "64QAM", 3, 8),
    (2048, 144, 1200, "QPSK", 3, 8), (2048, 144, 1200, "16QAM", 2, 12), (2048, 144, 1200, "64QAM", 2, 8),
    (4096, 288, 2400, "QPSK", 2, 4), (4096, 288, 2400, "16QAM", 2, 8), (4096, 288, 2400, "64QAM", 2, 4),
])
def test_batch_demod_vs_oracle(om, N, cp, Kd, mod, n_frames, n_sym):
    L = N + cp
    tail = cp + 5                                           # ragged: frame_len is not a multiple of L
    bits, iq = _frames(N, cp, Kd, n_sym, n_frames, mod, seed=N + n_sym, tail=tail)
    frame_len = iq.shape[1]
    bps = orc.BITS_PER_SYMBOL[mod]
    rx = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7, modulation=mod)
    nds = rx.data_symbols_per_frame(frame_len)
    assert nds == (n_sym // 4) * 3
    d_iq = om.DeviceBuffer(iq.nbytes).upload(iq)
    d_eq = poisoned(om, n_frames * nds * Kd * 8)
    d_bp = poisoned(om, n_frames * nds * Kd * bps // 8)
    d_bu = poisoned(om, n_frames * nds * Kd * bps)
    d_tsr = poisoned(om, n_frames * 16)
    assert rx.demod_frames(d_iq, n_frames, frame_len, frame_len, d_eq, d_bp, om.BITS_PACKED, d_tsr) == nds
    rx.demod_frames(d_iq, n_frames, frame_len, frame_len, None, d_bu, om.BITS_UNPACKED, None)
    eq = d_eq.download(np.complex64, n_frames * nds * Kd).reshape(n_frames, nds, Kd)
    bp = d_bp.download(np.uint8, n_frames * nds * Kd * bps // 8).reshape(n_frames, -1)
    bu = d_bu.download(np.uint8, n_frames * nds * Kd * bps).reshape(n_frames, -1)
    tsr = d_tsr.download(np.int32, n_frames * 4).reshape(n_frames, 4)
    rows = [r for r in range(n_sym) if r % 4 != 3]
    for f in range(n_frames):
        o = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, 100, 0.7, force_fp64=True)
        o.work(iq[f], np.zeros(frame_len, np.complex64))
        assert tsr[f, 0] == o.time_synch_ref[0] and tsr[f, 1] == o.time_synch_ref[1] and tsr[f, 3] == 1
        assert abs(tsr[f, 2] - o.time_synch_ref[2]) <= 1
        assert_close(eq[f], o.est_data_freq[rows], "equalised symbols of frame %d" % f)
        st = rx.frame_state(f)
        assert relerr(st["chan_freq"], o.est_chan_freq_P[0]) < TOL
        assert relerr(st["chan_time"], o.est_chan_time[0]) < TOL
        # bits: exact vs the transmitted bits and vs the oracle's de-map of the GPU's own equalised symbols
        assert np.array_equal(bu[f], orc.demap_hard(eq[f].ravel(), mod))
        assert np.array_equal(np.unpackbits(bp[f]), bu[f])
        if mod == "QPSK":
            assert np.array_equal(bu[f], bits[f])


def test_batch_frames_without_sync_and_short_frames(om):
    """Frame 1 is pure noise (no detection -> zeros, detected flag 0); frames shorter than a pattern give 0 symbols."""
    N, cp, Kd, n_sym = 64, 16, 60, 8
    bits, iq = _frames(N, cp, Kd, n_sym, 3, "QPSK", seed=1, tail=0)
    rng = np.random.default_rng(0)
    iq[1] = (0.01 * (rng.standard_normal(iq.shape[1]) + 1j * rng.standard_normal(iq.shape[1]))).astype(np.complex64)
    rx = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7)
    fl = iq.shape[1]
    nds = rx.data_symbols_per_frame(fl)
    d_iq = om.DeviceBuffer(iq.nbytes).upload(iq)
    d_eq = om.DeviceBuffer(3 * nds * Kd * 8)
    d_tsr = om.DeviceBuffer(3 * 16)
    rx.demod_frames(d_iq, 3, fl, fl, d_eq, None, om.BITS_NONE, d_tsr)
    eq = d_eq.download(np.complex64, 3 * nds * Kd).reshape(3, nds, Kd)
    tsr = d_tsr.download(np.int32, 12).reshape(3, 4)
    assert list(tsr[:, 3]) == [1, 0, 1] and not tsr[1].any()
    assert not eq[1].any() and eq[0].any() and eq[2].any()
    assert rx.demod_frames(d_iq, 3, fl, 3 * 80, d_eq, None, om.BITS_NONE, None) == 0      # < one [1,3] pattern
    assert rx.demod_frames(d_iq, 0, fl, fl, d_eq, None, om.BITS_NONE, None) == nds        # empty batch is a no-op
    with pytest.raises(ValueError):
        rx.demod_frames(d_iq, 3, fl - 1, fl, d_eq, None, om.BITS_NONE, None)              # stride < length


@pytest.mark.parametrize("N,cp,Kd,mod", [(64, 16, 60, "QPSK"), (64, 16, 60, "BPSK"), (256, 18, 152, "16QAM"),
                                         (1024, 72, 600, "64QAM"), (2048, 144, 1200, "QPSK"), (4096, 288, 2400, "16QAM"),
                                         # from 1024-pt up the symbol is stored from registers, the prefix from the register slots that
                                         # reach into the last cp samples: odd, one-sample, multi-slot and longer-than-N/2 prefixes
                                         (1024, 73, 600, "16QAM"), (4096, 1, 2400, "QPSK"), (2048, 700, 1200, "64QAM"),
                                         (2048, 1100, 1198, "16QAM"), (1024, 64, 1022, "QPSK")])
def test_tx_chain_vs_oracle(om, N, cp, Kd, mod):
    rng = np.random.default_rng(N + len(mod))
    n_sym, n_frames = 8, 3
    bps = orc.BITS_PER_SYMBOL[mod]
    tx = om.TxEngine(N, cp, N - 2, Kd, (1, 3), mod)
    nb = tx.bits_per_frame(n_sym)
    assert nb == 6 * Kd * bps
    bits = rng.integers(0, 2, (n_frames, nb)).astype(np.uint8)
    L = N + cp
    d_bits = om.DeviceBuffer(bits.nbytes).upload(bits)
    d_iq = om.DeviceBuffer(n_frames * n_sym * L * 8)
    tx.modulate_frames(d_bits, n_frames, n_sym, d_iq, bits_mode=om.BITS_UNPACKED)
    iq = d_iq.download(np.complex64, n_frames * n_sym * L).reshape(n_frames, -1)
    for f in range(n_frames):
        assert relerr(iq[f], orc.tx_modulate(bits[f], N, cp, N - 2, Kd, n_sym, modulation=mod)) < TOL
    if nb % 8 == 0:
        pk = np.packbits(bits, axis=1)
        d_pk = om.DeviceBuffer(pk.nbytes).upload(pk)
        d_iq2 = om.DeviceBuffer(n_frames * n_sym * L * 8)
        tx.modulate_frames(d_pk, n_frames, n_sym, d_iq2, bits_mode=om.BITS_PACKED)
        assert np.array_equal(d_iq2.download(np.complex64, n_frames * n_sym * L), iq.ravel())


def test_tx_chain_reproduces_reference_fixture(om, golden):
    """bits fixture -> HIP TX == reference tx_data_online fixture (the file the reference TX block replays)."""
    fx = golden("ref_fixtures.npz")
    tx = om.TxEngine(64, 16, 62, 60, (1, 3), "QPSK")
    bits = fx["tx_bits"][0].astype(np.uint8)
    d_bits = om.DeviceBuffer(bits.nbytes).upload(bits)
    d_iq = om.DeviceBuffer(19200 * 8)
    tx.modulate_frames(d_bits, 1, 240, d_iq)
    assert relerr(d_iq.download(np.complex64, 19200), fx["tx_online"][0]) < TOL


def test_channel_vs_oracle_and_noise_statistics(om, golden):
    fx = golden("ref_fixtures.npz")
    x = fx["tx_online"][0].astype(np.complex64)
    taps = (orc.REF_TAPS / np.linalg.norm(orc.REF_TAPS)).astype(np.complex64)
    tx = om.TxEngine(64, 16, 62, 60)
    d_x = om.DeviceBuffer(x.nbytes).upload(x)
    d_t = om.DeviceBuffer(taps.nbytes).upload(taps)
    n_out = len(x) + len(taps) - 1
    d_y = om.DeviceBuffer(n_out * 8)
    tx.channel(d_x, 1, len(x), len(x), d_t, len(taps), d_y, n_out, n_out)
    y = d_y.download(np.complex64, n_out)
    ref = orc.channel_apply(x, orc.REF_TAPS, 64)[:n_out]
    assert relerr(y, ref) < TOL
    assert np.max(np.abs(y - fx["tx_offline"][0][:n_out])) < 2e-4          # reference fixture carries 100 dB AWGN
    # AWGN: variance, zero mean, I/Q balance, reproducible from the seed, different per seed
    nv = 0.25
    tx.channel(d_x, 1, len(x), len(x), d_t, len(taps), d_y, n_out, n_out, noise_var=nv, seed=1234)
    n1 = d_y.download(np.complex64, n_out) - y
    tx.channel(d_x, 1, len(x), len(x), d_t, len(taps), d_y, n_out, n_out, noise_var=nv, seed=1234)
    assert np.array_equal(d_y.download(np.complex64, n_out) - y, n1)
    tx.channel(d_x, 1, len(x), len(x), d_t, len(taps), d_y, n_out, n_out, noise_var=nv, seed=99)
    assert not np.array_equal(d_y.download(np.complex64, n_out) - y, n1)
    assert abs(np.var(n1) / nv - 1) < 0.05 and abs(np.mean(n1)) < 0.02
    assert abs(np.var(n1.real) / np.var(n1.imag) - 1) < 0.1


def test_demap_vs_reference_bitrecovery(om, golden):
    import OFDMReceiver
    g = golden("ref_bitrecovery.npz")
    blk = OFDMReceiver.BitRecovery("QPSK", "/tmp/", 0)
    assert blk.work([g["z"]], [None]) == len(g["z"])
    # every symbol, including the four planted on an axis / at the origin (gen_golden.py): there the reference's nearest-point
    # search is a tie that its fp64 arithmetic decides, and the device repeats that arithmetic literally (qpsk_bits_on_axis)
    assert g["z"][0].real == 0 and g["z"][1].imag == 0 and g["z"][2] == 0 and g["z"][3].imag == 0
    assert np.array_equal(blk.hardbit.ravel(), g["hardbit"])
    assert relerr(blk.softbit0, g["softbit0"]) < TOL
    assert relerr(blk.softbit1, g["softbit1"]) < TOL
    # outlier flip + all modulations, hard decisions vs the oracle
    rng = np.random.default_rng(2)
    z = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
    rx = om.RxEngine(1, 64, 16, 62, (1, 3), 60, 100)
    d_z = om.DeviceBuffer(z.nbytes).upload(z)
    for mod in ("BPSK", "QPSK", "16QAM", "64QAM"):
        bps = orc.BITS_PER_SYMBOL[mod]
        d_h = om.DeviceBuffer(len(z) * bps)
        rx.demap(d_z, len(z), mod, d_h)
        om.load().ofdm_device_synchronize(0)
        assert np.array_equal(d_h.download(np.uint8, len(z) * bps), orc.demap_hard(z, mod))
    # many ties: symbols on either axis (incl. +-0, the outlier edges and the origin) against the literal restatement
    v = np.concatenate([rng.standard_normal(3000), [0.0, -0.0, 1.41421354, -1.41421354, 1.4142137, 0.70710677, -0.70710677, 1e-30, 3.5]])
    v = v.astype(np.float32)
    zt = np.concatenate([1j * v, v + 0j, -0.0 + 1j * v, v - 0.0j]).astype(np.complex64)
    blk2 = OFDMReceiver.BitRecovery("QPSK", "/tmp/", 0)
    blk2.work([zt], [None])
    assert np.array_equal(blk2.hardbit.ravel(), orc.bit_recovery(zt)[0])
    assert np.array_equal(blk2.hardbit.ravel(), orc.demap_hard(zt, "QPSK"))


def test_fused_demapper_on_exact_zero_symbols(om):
    """Data bins outside the sync span have no channel estimate: H = 0, gain = 0, the equalised symbol is exactly 0+0j -- the
    four-way tie of BitRecovery's nearest-point search.  The fused de-mapper (packed and unpacked) must give the reference's
    bits there too: (1, 0)."""
    N, cp, Ks, Kd, n_sym, n_frames = 64, 16, 30, 60, 8, 3
    rng = np.random.default_rng(3)
    nb = 6 * Kd * 2
    bits = rng.integers(0, 2, (n_frames, nb)).astype(np.uint8)
    iq = np.stack([orc.tx_modulate(bits[f], N, cp, Ks, Kd, n_sym) for f in range(n_frames)]).astype(np.complex64)
    fl = iq.shape[1]
    rx = om.RxEngine(n_sym, N, cp, Ks, (1, 3), Kd, 100, 0.7)
    d_iq = om.DeviceBuffer(iq.nbytes).upload(iq)
    d_eq = om.DeviceBuffer(n_frames * 6 * Kd * 8)
    d_bp = om.DeviceBuffer(n_frames * 6 * Kd * 2 // 8)
    d_bu = om.DeviceBuffer(n_frames * 6 * Kd * 2)
    rx.demod_frames(d_iq, n_frames, fl, fl, d_eq, d_bp, om.BITS_PACKED, None)
    rx.demod_frames(d_iq, n_frames, fl, fl, None, d_bu, om.BITS_UNPACKED, None)
    eq = d_eq.download(np.complex64, n_frames * 6 * Kd).reshape(n_frames, 6, Kd)
    bu = d_bu.download(np.uint8, n_frames * 6 * Kd * 2)
    bp = d_bp.download(np.uint8, n_frames * 6 * Kd * 2 // 8)
    outside = np.r_[0:15, 45:60]                           # list entries of bins |k| > Ks/2
    assert not eq[:, :, outside].any() and eq[:, :, 15:45].all()
    want = orc.bit_recovery(eq.ravel())[0]
    assert np.array_equal(bu, want) and np.array_equal(np.unpackbits(bp), want)
    assert np.array_equal(want.reshape(n_frames, 6, Kd, 2)[:, :, outside], np.broadcast_to([1, 0], (n_frames, 6, 30, 2)))
    inside = bu.reshape(n_frames, 6, Kd, 2)[:, :, 15:45]
    assert np.array_equal(inside, bits.reshape(n_frames, 6, Kd, 2)[:, :, 15:45])


def test_loopback_property_full_size_numerology(om):
    """2048/144/1200 at a size the oracle would need minutes for: bits -> HIP TX -> HIP channel(+AWGN 30 dB)
    -> HIP RX -> bits is the identity; and the equaliser output is invariant to a common input gain."""
    N, cp, Kd, n_sym, n_frames = 2048, 144, 1200, 240, 12
    L = N + cp
    rng = np.random.default_rng(0)
    txe = om.TxEngine(N, cp, N - 2, Kd, (1, 3), "QPSK")
    rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 30, 0.7)
    nb = txe.bits_per_frame(n_sym)
    bits = rng.integers(0, 2, (n_frames, nb)).astype(np.uint8)
    d_bits = om.DeviceBuffer(bits.nbytes).upload(bits)
    fl = n_sym * L
    d_tx = om.DeviceBuffer(n_frames * fl * 8)
    d_rx = om.DeviceBuffer(n_frames * fl * 8)
    taps = (orc.REF_TAPS / np.linalg.norm(orc.REF_TAPS)).astype(np.complex64)
    d_t = om.DeviceBuffer(taps.nbytes).upload(taps)
    txe.modulate_frames(d_bits, n_frames, n_sym, d_tx)
    txe.channel(d_tx, n_frames, fl, fl, d_t, len(taps), d_rx, fl, fl, noise_var=1e-3, seed=7)
    nds = rxe.data_symbols_per_frame(fl)
    d_out = om.DeviceBuffer(n_frames * nds * Kd * 2)
    d_eq = om.DeviceBuffer(n_frames * nds * Kd * 8)
    assert rxe.demod_frames(d_rx, n_frames, fl, fl, d_eq, d_out, om.BITS_UNPACKED, None) == 180
    got = d_out.download(np.uint8, n_frames * nds * Kd * 2).reshape(n_frames, -1)
    assert np.array_equal(got, bits)
    eq1 = d_eq.download(np.complex64, n_frames * nds * Kd)
    # scale invariance (per-symbol power normalisation on both sync and data paths)
    x = d_rx.download(np.complex64, n_frames * fl)
    d_rx.upload((x * np.complex64(3.0)).astype(np.complex64))
    rxe.demod_frames(d_rx, n_frames, fl, fl, d_eq, None, om.BITS_NONE, None)
    assert relerr(d_eq.download(np.complex64, n_frames * nds * Kd), eq1) < TOL


def test_offline_modulator_mirror_reproduces_fixture(om, golden):
    """The reference's SDRScript flow through the mirrored classes: bits fixture -> tx_data_online / tx_data_offline."""
    from ofdm_mi355x.txrx_mod import OFDM, MultiAntennaSystem, SynchSignal
    fx = golden("ref_fixtures.npz")
    N, cp, Kd, n_sym = 64, 16, 60, 240
    all_bins = np.array(list(range(-Kd // 2, 0)) + list(range(1, Kd // 2 + 1)))
    pat = np.tile(np.array([0, 1, 1, 1]), n_sym // 4)
    mas = MultiAntennaSystem(OFDM(cp, Kd, "QPSK", N, 15e3), 1, "SpMult", all_bins, n_sym, pat, 15e3 * N, "LTE-TU", 0, "Fading",
                             1, all_bins, np.array([], dtype=int))
    caz = SynchSignal(cp, N - 2, 1, N, np.array([1, 3]))
    assert relerr(caz.ZChu0, orc.zadoff_chu(62, 23)) < 1e-12
    mas.multi_ant_binary_map(caz, fx["tx_bits"], np.array([1, 3]))
    mas.multi_ant_symb_gen(n_sym)
    assert relerr(mas.buffer_data_tx_time[0], fx["tx_online"][0]) < TOL
    mas.rx_signal_gen()
    assert mas.buffer_data_rx_time.shape == fx["tx_offline"].shape
    assert np.max(np.abs(mas.buffer_data_rx_time[0] - fx["tx_offline"][0])) < 2e-4       # fixture carries 100 dB AWGN
    mas.additive_noise("Digital", 20, "Fading", "Complex")
    noise = mas.buffer_data_rx_time[0] - orc.channel_apply(mas.buffer_data_tx_time[0], orc.REF_TAPS, N)
    nz = noise[:mas._noise_len]
    assert abs(np.var(nz) / mas.noise_var - 1) < 0.1 and not noise[mas._noise_len:].any()


def test_batch_path_is_graph_capturable_and_stream_ordered(om):
    """ofdm_rx_demod_frames allocates nothing after ofdm_rx_reserve and only enqueues on the caller's stream: it can be
    captured into a hipGraph (torch.cuda.CUDAGraph) and replayed; results equal the eager call."""
    import torch
    N, cp, Kd, n_sym, n_frames = 1024, 72, 600, 8, 6
    bits, iq = _frames(N, cp, Kd, n_sym, n_frames, "QPSK", seed=5, tail=0)
    fl = iq.shape[1]
    rx = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7)
    rx.reserve(n_frames)
    nds = rx.data_symbols_per_frame(fl)
    d_iq = torch.from_numpy(iq.view(np.float32).reshape(n_frames, fl, 2)).cuda()
    d_eq = torch.zeros((n_frames, nds, Kd, 2), dtype=torch.float32, device="cuda")
    d_b = torch.zeros((n_frames, nds * Kd * 2 // 8), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        rx.demod_frames(d_iq, n_frames, fl, fl, d_eq, d_b, om.BITS_PACKED, None, s.cuda_stream)
    s.synchronize()
    eager_eq, eager_b = d_eq.clone(), d_b.clone()
    assert np.array_equal(np.unpackbits(eager_b.cpu().numpy(), axis=1), bits)
    d_eq.zero_()
    d_b.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rx.demod_frames(d_iq, n_frames, fl, fl, d_eq, d_b, om.BITS_PACKED, None, torch.cuda.current_stream().cuda_stream)
    assert not d_eq.any()                      # capture enqueues nothing
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(d_eq, eager_eq) and torch.equal(d_b, eager_b)


def test_two_handles_run_concurrently_from_threads(om):
    """GNU Radio runs one thread per block: two handles, two host threads, no shared mutable state in the library."""
    import threading
    N, cp, Kd, n_sym = 256, 18, 152, 8
    results = {}

    def run(idx):
        bits, iq = _frames(N, cp, Kd, n_sym, 3, "QPSK", seed=100 + idx, tail=0)
        fl = iq.shape[1]
        rx = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7)
        d_iq = om.DeviceBuffer(iq.nbytes).upload(iq)
        nds = rx.data_symbols_per_frame(fl)
        d_b = om.DeviceBuffer(3 * nds * Kd * 2)
        ok = True
        for _ in range(20):
            rx.demod_frames(d_iq, 3, fl, fl, None, d_b, om.BITS_UNPACKED, None)
            ok &= np.array_equal(d_b.download(np.uint8, 3 * nds * Kd * 2).reshape(3, -1), bits)
        results[idx] = ok

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert results == {0: True, 1: True}


def _decision_margin(z, mod):
    """Distance of every symbol to the nearest decision boundary of `demap_hard` (per axis, the smaller of the two)."""
    z = np.asarray(z).ravel()
    if mod == "QPSK":
        edges = np.array([0.0, np.sqrt(2.0), -np.sqrt(2.0)])
    elif mod == "16QAM":
        edges = np.array([0.0, 2, -2]) / np.sqrt(10.0)
    else:
        edges = np.array([0.0, 2, -2, 4, -4, 6, -6]) / np.sqrt(42.0)
    dr = np.min(np.abs(z.real[:, None] - edges[None, :]), axis=1)
    di = np.min(np.abs(z.imag[:, None] - edges[None, :]), axis=1)
    return np.minimum(dr, di)


CFG3_BER_SEED99 = 0.041454   # gpurun_out/r03y/fullsize.log: 4369 frames, seed 99, 64-QAM, per-frame Rayleigh taps, 30 dB


@pytest.mark.parametrize("config", ["cfg2", "cfg3"])
def test_full_size_batch_properties(om, config):
    """BASELINE.json's full sizes (config 2: 4369 frames x 240 symbols = 1,048,560 symbols, 16-QAM, AWGN; config 3: 64-QAM
    under per-frame Rayleigh taps), far beyond what the oracle can process, checked through size-independent properties:
    (i) config 2: bits -> HIP TX -> HIP channel -> HIP RX -> bits is the identity on ALL 3.8e9 bits;
    (ii) determinism: a second run reproduces every output byte; (iii) frames are independent: demodulating the batch in
    reversed frame order returns the reversed outputs bit for bit; (iv) the equalised symbols are invariant to a common
    (power-of-two) gain on the input (per-symbol power normalisation); (v) a checksum of per-frame checksums ties (ii)-(iii) together."""
    import torch
    import bench
    cfg = dict(bench.CONFIGS[config])
    n_frames = cfg["frames"]                                   # both configs at BASELINE's 4369 frames (1,048,560 symbols)
    N, cp, Kd, mod, n_sym = cfg["nfft"], cfg["cp"], cfg["Kd"], cfg["mod"], cfg["n_sym"]
    fl = n_sym * (N + cp)
    torch.cuda.set_device(0)
    # a real torch stream: handle 0 (torch's default stream) means "the library's own stream" to the C ABI, which would leave the
    # kernels unordered with the torch copies below
    torch.cuda.set_stream(torch.cuda.Stream())
    d_rx, tx_bits, _ = bench.build_inputs(torch, om, cfg, n_frames, 0, seed=99)
    rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, cfg["snr_db"], cfg.get("gate", 0.7), modulation=mod, device=0)
    rxe.reserve(n_frames)
    rxe.set_max_trials(N + cp)
    nds = rxe.data_symbols_per_frame(fl)
    nbytes = nds * Kd * bench.BPS[mod] // 8
    st = torch.cuda.current_stream().cuda_stream

    def run(src):
        eq = torch.empty((n_frames, nds, Kd, 2), dtype=torch.float32, device="cuda")
        bits = torch.empty((n_frames, nbytes), dtype=torch.uint8, device="cuda")
        assert rxe.demod_frames(src, n_frames, fl, fl, eq, bits, om.BITS_PACKED, None, st) == nds
        torch.cuda.synchronize()
        return eq, bits

    eq1, b1 = run(d_rx)
    # sampled frames against the oracle: first, last and two seeded random frames of the full-size batch
    sample = sorted({0, n_frames - 1, *np.random.default_rng(5).integers(1, n_frames - 1, 2).tolist()})
    rows = [r for r in range(n_sym) if r % 4 != 3]
    for f in sample:
        iq_f = d_rx[f].cpu().numpy().view(np.complex64).ravel()
        o = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, cfg["snr_db"], cfg.get("gate", 0.7), force_fp64=True)
        o.work(iq_f, np.zeros(fl, np.complex64))
        ref = o.est_data_freq[rows]
        assert o.time_synch_ref[0] == cp and rxe.frame_state(f)["chan_freq"].any()
        assert_close(eq1[f].cpu().numpy().view(np.complex64).reshape(nds, Kd), ref, "%s frame %d" % (config, f))
        got_bits = np.unpackbits(b1[f].cpu().numpy())
        want_bits = orc.demap_hard(ref.ravel(), mod)
        safe = np.repeat(_decision_margin(ref.ravel(), mod) > 1e-4, bench.BPS[mod])
        assert safe.mean() > 0.99 and np.array_equal(got_bits[safe], want_bits[safe])
    if config == "cfg2":
        assert torch.equal(b1, tx_bits)                                    # (i) 3.77e9 bits, zero errors
    else:
        lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32, device="cuda")
        nerr = sum(int(lut[(b1[f0:f0 + 128] ^ tx_bits[f0:f0 + 128]).long()].sum().item()) for f0 in range(0, n_frames, 128))
        # 64-QAM under 8-tap Rayleigh fading with a one-tap equaliser does err in the faded bins.  Inputs, taps and noise are seeded
        # and every kernel is deterministic, so the batch's bit error rate is a fixed number: pinned to the value of the first run
        # of this test (round 3), with room only for a different chip's transcendentals in the noise generator.
        ber = nerr / (b1.numel() * 8)
        print("config 3 bit error rate over %d frames: %.6f" % (n_frames, ber))
        assert abs(ber - CFG3_BER_SEED99) < 3e-4, ber
    eq2, b2 = run(d_rx)
    assert torch.equal(b1, b2) and torch.equal(eq1, eq2)                   # (ii)
    w = torch.arange(1, nbytes + 1, device="cuda", dtype=torch.int64) % 251
    frame_sums = (b1.to(torch.int64) * w).sum(dim=1)                      # (v) per-frame checksums
    total = int((frame_sums * (torch.arange(n_frames, device="cuda") % 65521 + 1)).sum().item())
    del eq2, b2
    rev = torch.empty_like(d_rx)                   # frame order reversed, in slices (one flip over 4.6e9 elements is avoided)
    for f0 in range(0, n_frames, 256):
        nf = min(256, n_frames - f0)
        rev[n_frames - f0 - nf:n_frames - f0] = torch.flip(d_rx[f0:f0 + nf], dims=[0])
    eq3, b3 = run(rev)
    # (iii) row by row (one flip of a 4369 x 108000 uint8 tensor returned zero rows with this torch build; rows are compared directly)
    bad_b = [i for i in range(n_frames) if not torch.equal(b3[n_frames - 1 - i], b1[i])]
    bad_e = [i for i in range(n_frames) if not torch.equal(eq3[n_frames - 1 - i], eq1[i])]
    assert not bad_b and not bad_e, (len(bad_b), bad_b[:8], len(bad_e), bad_e[:8])
    fs3 = (b3.to(torch.int64) * w).sum(dim=1)
    wr = (n_frames - 1 - torch.arange(n_frames, device="cuda")) % 65521 + 1       # weight of the ORIGINAL frame index
    assert int((fs3 * wr).sum().item()) == total
    del eq3, b3, rev
    d_rx.mul_(4.0)                                   # (iv) a power of two: exact in fp32, so even boundary symbols keep their bits
    eq4, b4 = run(d_rx)
    assert torch.equal(b4, b1)
    num = float((eq4 - eq1).abs().max().item())
    den = float(eq1.abs().max().item())
    assert num / den < TOL


@pytest.mark.parametrize("n,off_a,off_b", [(0, 0, 0), (1, 0, 0), (15, 0, 0), (16, 0, 0), (17, 0, 0), (4097, 0, 0), ((1 << 20) + 3, 0, 0),
                                            (70001, 1, 0), (70001, 3, 7), (1 << 16, 16, 32)])
def test_count_bit_errors_equals_numpy(om, n, off_a, off_b):
    """popcount(a ^ b) over two device byte strings (the multi-GPU BER numerator): exact for every length, tail and
    alignment; the counter accumulates across calls."""
    import torch
    rng = np.random.default_rng(n + off_a)
    a = rng.integers(0, 256, n + 64, dtype=np.uint8)
    b = a.copy()
    flip = rng.random(n + 64) < 0.1
    b[flip] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    want = int(np.unpackbits(a[off_a:off_a + n] ^ b[off_b:off_b + n]).sum())
    om.count_bit_errors(da.data_ptr() + off_a, db.data_ptr() + off_b, n, cnt.data_ptr())
    torch.cuda.synchronize()
    assert int(cnt.item()) == want
    om.count_bit_errors(da.data_ptr() + off_a, db.data_ptr() + off_b, n, cnt.data_ptr())
    torch.cuda.synchronize()
    assert int(cnt.item()) == 2 * want


@pytest.mark.parametrize("N,cp,Ks,Kd,mod", [(1024, 72, 1022, 602, "QPSK"), (2048, 144, 2046, 1198, "16QAM"), (4096, 288, 4094, 2402, "64QAM"),
                                             (1024, 72, 1024, 1024, "16QAM"),
                                             # list edges ON register-slot / 128-entry block boundaries, nearly empty and nearly full lists:
                                             # the scalar "slot wholly listed / wholly unlisted / block inside Kd" decisions of the kernel
                                             (2048, 144, 2046, 1024, "16QAM"), (2048, 144, 2046, 256, "QPSK"), (2048, 144, 2046, 2044, "64QAM"),
                                             (2048, 144, 2046, 1280, "16QAM"), (1024, 72, 1022, 128, "16QAM"), (1024, 72, 1022, 1020, "QPSK"),
                                             (4096, 288, 4094, 512, "16QAM"), (4096, 288, 4094, 4092, "QPSK"), (4096, 288, 4094, 2048, "64QAM")])
def test_dense_output_mapping_on_awkward_bin_counts(om, N, cp, Ks, Kd, mod):
    """From 1024-pt up a lane of the demod kernel owns two PAIRS of list entries 128 apart and bits leave in groups of four
    entries assembled across a lane pair.  Bin counts that are even but not a multiple of 4 (the last pair of a row has no
    partner: equalised symbols and one-bit-per-byte output only -- packed output is refused for them, as before), and K == N
    (bin N/2 listed twice, every list entry used), against the fp64 oracle."""
    n_sym, n_frames = 8, 2
    L = N + cp
    bps = orc.BITS_PER_SYMBOL[mod]
    rng = np.random.default_rng(Kd)
    nb = (n_sym // 4) * 3 * Kd * bps
    iq = np.zeros((n_frames, n_sym * L + cp + 3), np.complex64)
    for f in range(n_frames):
        tx = orc.tx_modulate(rng.integers(0, 2, nb), N, cp, Ks, Kd, n_sym, modulation=mod)
        iq[f] = orc.channel_apply(tx, orc.REF_TAPS, N)[:iq.shape[1]]
    fl = iq.shape[1]
    rx = om.RxEngine(n_sym, N, cp, Ks, (1, 3), Kd, 100, 0.7, modulation=mod)
    nds = rx.data_symbols_per_frame(fl)
    d_iq = om.DeviceBuffer(iq.nbytes).upload(iq)
    d_eq = om.DeviceBuffer(n_frames * nds * Kd * 8)
    d_bu = om.DeviceBuffer(n_frames * nds * Kd * bps)
    assert rx.demod_frames(d_iq, n_frames, fl, fl, d_eq, d_bu, om.BITS_UNPACKED, None) == nds
    eq = d_eq.download(np.complex64, n_frames * nds * Kd).reshape(n_frames, nds, Kd)
    bu = d_bu.download(np.uint8, n_frames * nds * Kd * bps).reshape(n_frames, -1)
    rows = [r for r in range(n_sym) if r % 4 != 3]
    for f in range(n_frames):
        o = orc.RxOracle(n_sym, N, cp, Ks, [1, 3], Kd, 100, 0.7, force_fp64=True)
        o.work(iq[f], np.zeros(fl, np.complex64))
        assert_close(eq[f], o.est_data_freq[rows], "frame %d" % f)
        assert np.array_equal(bu[f], orc.demap_hard(eq[f].ravel(), mod))
    if Kd % 4 == 0:
        d_bp = om.DeviceBuffer(n_frames * nds * Kd * bps // 8)
        rx.demod_frames(d_iq, n_frames, fl, fl, None, d_bp, om.BITS_PACKED, None)
        assert np.array_equal(np.unpackbits(d_bp.download(np.uint8, n_frames * nds * Kd * bps // 8)), bu.ravel())
    else:
        with pytest.raises(ValueError):
            rx.demod_frames(d_iq, n_frames, fl, fl, None, om.DeviceBuffer(n_frames * nds * Kd * bps // 8 + 8), om.BITS_PACKED, None)


def test_lds_request_grows_with_the_bin_count_inside_one_process(om):
    """The demod kernel's LDS request includes the per-frame gain table (Kd entries): at 4096-pt it crosses 64 KB, which has to be
    announced per kernel -- and announced AGAIN when a later engine asks for more (ascending Kd, same instantiation)."""
    N, cp, Ks, mod, n_sym = 4096, 288, 4094, "16QAM", 8
    L = N + cp
    for Kd in (3600, 3840, 4092):
        rng = np.random.default_rng(Kd)
        tx = orc.tx_modulate(rng.integers(0, 2, (n_sym // 4) * 3 * Kd * 4), N, cp, Ks, Kd, n_sym, modulation=mod)
        iq = orc.channel_apply(tx, orc.REF_TAPS, N)[:n_sym * L + cp + 3][None, :].astype(np.complex64)
        fl = iq.shape[1]
        rx = om.RxEngine(n_sym, N, cp, Ks, (1, 3), Kd, 100, 0.7, modulation=mod)
        nds = rx.data_symbols_per_frame(fl)
        d_iq = om.DeviceBuffer(iq.nbytes).upload(iq)
        d_eq = om.DeviceBuffer(nds * Kd * 8)
        d_bp = om.DeviceBuffer(nds * Kd * 4 // 8)
        assert rx.demod_frames(d_iq, 1, fl, fl, d_eq, d_bp, om.BITS_PACKED, None) == nds
        eq = d_eq.download(np.complex64, nds * Kd).reshape(nds, Kd)
        o = orc.RxOracle(n_sym, N, cp, Ks, [1, 3], Kd, 100, 0.7, force_fp64=True)
        o.work(iq[0], np.zeros(fl, np.complex64))
        assert_close(eq, o.est_data_freq[[r for r in range(n_sym) if r % 4 != 3]], "Kd %d" % Kd)
        assert np.array_equal(np.unpackbits(d_bp.download(np.uint8, nds * Kd * 4 // 8)), orc.demap_hard(eq.ravel(), mod))


@pytest.mark.parametrize("N,cp,Kd,mod,n_frames", [(64, 16, 60, "QPSK", 6000), (128, 9, 72, "16QAM", 3000), (256, 18, 180, "16QAM", 1500),
                                                  (512, 36, 300, "64QAM", 1200)])
def test_small_sizes_many_frames_loopback_and_sampled_oracle(om, N, cp, Kd, mod, n_frames):
    """Below 1024-pt a chunk of the demod launch is a whole frame or a large part of one once the batch is big (few-frame
    batches, which every other small-size test uses, are cut finer): bits -> HIP TX -> HIP channel -> HIP RX is the identity on
    EVERY frame of a large batch (packed bits, device-side error count), and sampled frames equal the oracle's rows."""
    import torch
    n_sym, L = 240, N + cp
    bps = orc.BITS_PER_SYMBOL[mod]
    txe = om.TxEngine(N, cp, N - 2, Kd, (1, 3), mod)
    rxe = om.RxEngine(n_sym, N, cp, N - 2, (1, 3), Kd, 100, 0.7, modulation=mod)
    nb = txe.bits_per_frame(n_sym)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(N)
    bits = torch.randint(0, 256, (n_frames, nb // 8), dtype=torch.uint8, device=dev, generator=gen)
    fl = n_sym * L
    iq = torch.empty((n_frames, fl, 2), dtype=torch.float32, device=dev)
    rxb = torch.empty_like(iq)
    taps = torch.zeros((1, 2), dtype=torch.float32, device=dev)
    taps[0, 0] = 1
    txe.modulate_frames(bits.data_ptr(), n_frames, n_sym, iq.data_ptr(), bits_mode=om.BITS_PACKED)
    txe.channel(iq.data_ptr(), n_frames, fl, fl, taps.data_ptr(), 1, rxb.data_ptr(), fl, fl, noise_var=1e-6, seed=3)
    nds = rxe.data_symbols_per_frame(fl)
    assert nds * Kd * bps == nb
    out = torch.full((n_frames, nb // 8), 0xA5, dtype=torch.uint8, device=dev)
    eq = torch.full((n_frames, nds * Kd, 2), float("nan"), dtype=torch.float32, device=dev)
    assert rxe.demod_frames(rxb.data_ptr(), n_frames, fl, fl, eq.data_ptr(), out.data_ptr(), om.BITS_PACKED, None) == nds
    torch.cuda.synchronize()
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    om.count_bit_errors(out.data_ptr(), bits.data_ptr(), out.numel(), cnt.data_ptr())
    torch.cuda.synchronize()
    assert int(cnt.item()) == 0
    rows = [r for r in range(n_sym) if r % 4 != 3]        # est_data_freq rows the patterns write (row p*(S+D) + n)
    for f in (0, n_frames // 2 + 1, n_frames - 1):
        x = torch.view_as_complex(rxb[f]).cpu().numpy()
        o = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, 100, 0.7, force_fp64=True)
        o.work(x, np.zeros(fl, np.complex64))
        got = torch.view_as_complex(eq[f]).cpu().numpy().reshape(nds, Kd)
        assert_close(got, o.est_data_freq[rows], "frame %d" % f)
