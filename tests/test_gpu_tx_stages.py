"""GPU parity of the decomposed transmitter (run with -m gpu).

The reference holds no code for these stages, so they are pinned to what it does hold: chained, the stage kernels must equal
the fused `tx_modulate_kernel` BIT FOR BIT (they share their device functions), which is itself checked against the oracle
and the reference's `tx_data_online` fixture; the chain is also compared with that fixture directly (1e-5) and every stage
with its oracle counterpart.  Pilots and the counter-based bit source have no reference counterpart: parity unpinned, checked
against the oracle's definition (the bit source bit-exactly, down to the Philox known-answer vectors)."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ofdm_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def om():
    import ofdm_mi355x
    ofdm_mi355x.load()
    return ofdm_mi355x


def _stages(om, bits, N, cp, Ks, Kd, mod, n_data, root=23, every=3, pilots=(), pv=1.0, packed=False):
    """bits -> map -> grid -> IFFT -> CP -> mux through the C ABI, every intermediate downloaded."""
    L = N + cp
    tx = om.TxEngine(N, cp, Ks, Kd, (1, every), mod, zc_root=root)
    if len(pilots):
        tx.set_pilots(pilots, pv)
    src = np.packbits(bits) if packed else bits
    d_bits = om.DeviceBuffer(max(src.nbytes, 8)).upload(src)
    n_sym = n_data * Kd
    d_sym = om.DeviceBuffer(n_sym * 8)
    tx.map(d_bits, n_sym, d_sym, om.BITS_PACKED if packed else om.BITS_UNPACKED)
    d_grid = om.DeviceBuffer(n_data * N * 8)
    tx.grid(d_sym, n_data, d_grid)
    d_time = om.DeviceBuffer(n_data * N * 8)
    tx.ifft_cp(d_grid, n_data, d_time, do_ifft=True, add_cp=False)
    d_cp = om.DeviceBuffer(n_data * L * 8)
    tx.ifft_cp(d_time, n_data, d_cp, do_ifft=False, add_cp=True)
    n_out = tx.mux_symbols(n_data)
    d_out = om.DeviceBuffer(n_out * L * 8)
    assert tx.mux(d_cp, n_data, d_out) == n_out
    d_one = om.DeviceBuffer(n_data * L * 8)
    tx.ifft_cp(d_grid, n_data, d_one, do_ifft=True, add_cp=True)
    return dict(sym=d_sym.download(np.complex64, n_sym), grid=d_grid.download(np.complex64, n_data * N).reshape(n_data, N),
                time=d_time.download(np.complex64, n_data * N).reshape(n_data, N),
                cp=d_cp.download(np.complex64, n_data * L).reshape(n_data, L),
                fused_time=d_one.download(np.complex64, n_data * L).reshape(n_data, L),
                out=d_out.download(np.complex64, n_out * L), tx=tx, n_out=n_out)


@pytest.mark.parametrize("N,cp,Kd,mod,n_sym", [
    (64, 16, 60, "QPSK", 24), (64, 16, 60, "BPSK", 8), (128, 32, 100, "16QAM", 12), (256, 64, 180, "64QAM", 8),
    (512, 36, 300, "QPSK", 8), (1024, 72, 600, "16QAM", 8), (2048, 144, 1200, "64QAM", 8), (4096, 288, 2400, "16QAM", 4),
    (64, 16, 64, "QPSK", 8),        # K == N: bin N/2 listed twice, the later entry wins in both paths
])
def test_chained_stages_equal_the_fused_kernel_bit_for_bit(om, N, cp, Kd, mod, n_sym):
    rng = np.random.default_rng(N + n_sym)
    bps = orc.BITS_PER_SYMBOL[mod]
    n_data = (n_sym // 4) * 3
    bits = rng.integers(0, 2, n_data * Kd * bps).astype(np.uint8)
    Ks = N - 2 if Kd < N else N
    st = _stages(om, bits, N, cp, Ks, Kd, mod, n_data)
    L = N + cp
    assert st["n_out"] == n_sym
    fused = om.TxEngine(N, cp, Ks, Kd, (1, 3), mod)
    d_bits = om.DeviceBuffer(bits.nbytes).upload(bits)
    d_iq = om.DeviceBuffer(n_sym * L * 8)
    fused.modulate_frames(d_bits, 1, n_sym, d_iq, bits_mode=om.BITS_UNPACKED)
    ref_gpu = d_iq.download(np.complex64, n_sym * L)
    assert np.array_equal(st["out"], ref_gpu)                                   # bit for bit
    assert np.array_equal(st["cp"], st["fused_time"])                           # IFFT and CP in one launch == two launches
    # every stage against its oracle counterpart
    sym = orc.map_bits(bits, mod)
    if mod in ("BPSK", "QPSK"):
        assert np.array_equal(st["sym"], sym.astype(np.complex64))
    assert relerr(st["sym"], sym) < 2e-7                     # 16/64-QAM levels are formed in fp32 on the device (1 ulp)
    if Kd < N:
        g = orc.tx_stage_grid(sym, N, Kd)
        assert np.array_equal(st["grid"] != 0, g != 0) and relerr(st["grid"], g) < 2e-7
        assert np.array_equal(st["grid"][:, orc.bins_p(Kd, N)].ravel(), st["sym"])
        assert relerr(st["time"], orc.tx_stage_ifft(g)) < TOL
        assert relerr(st["cp"], orc.tx_stage_cp(orc.tx_stage_ifft(g), cp)) < TOL
        assert relerr(st["out"], orc.tx_modulate(bits, N, cp, Ks, Kd, n_sym, modulation=mod)) < TOL
    if (n_data * Kd * bps) % 8 == 0:
        assert np.array_equal(_stages(om, bits, N, cp, Ks, Kd, mod, n_data, packed=True)["out"], st["out"])


def test_chain_reproduces_the_reference_fixture(om, golden):
    fx = golden("ref_fixtures.npz")
    bits = fx["tx_bits"][0].astype(np.uint8)
    st = _stages(om, bits, 64, 16, 62, 60, "QPSK", 180)
    assert st["n_out"] == 240
    assert relerr(st["out"], fx["tx_online"][0]) < TOL


def test_pilots_and_flowgraph_parameters_vs_oracle(om):
    """The flowgraph's own numbers (RXtransmit_6.grc: pilot_locations [-21,-7,7,21], prime_no 47, synch_every 3,
    synch_length fft-2, cp fft/4) at fft 64; plus a partial last group (5 data symbols)."""
    rng = np.random.default_rng(5)
    N, cp, Kd, pil = 64, 16, 58, [-21, -7, 7, 21]
    n_data = 5
    bits = rng.integers(0, 2, n_data * Kd * 2).astype(np.uint8)
    st = _stages(om, bits, N, cp, N - 2, Kd, "QPSK", n_data, root=47, every=3, pilots=pil, pv=1 - 1j)
    g = orc.tx_stage_grid(orc.map_bits(bits, "QPSK"), N, Kd, pil, 1 - 1j)
    assert np.array_equal(st["grid"], g.astype(np.complex64))
    ref = orc.tx_stage_mux(orc.tx_stage_cp(orc.tx_stage_ifft(g), cp), N, cp, 47, 3, N - 2)
    assert st["n_out"] == 7 == len(ref)
    assert relerr(st["out"], ref.ravel()) < TOL
    assert relerr(st["tx"].sync_symbol()[0], ref[0]) < TOL
    with pytest.raises(ValueError):
        st["tx"].set_pilots([0])                     # DC is not an occupied bin
    with pytest.raises(ValueError):
        st["tx"].set_pilots([7, 7])
    with pytest.raises(ValueError):
        st["tx"].set_pilots([40])                    # outside the occupied span


def test_mux_with_a_pattern_longer_than_one_launch_dimension(om):
    """`synch_every` comes straight from the flowgraph: S + D beyond 65535 (the blockIdx.y range -- the old launcher then computed
    a launch size of 0 and never terminated) and more output symbols than 65535 must both work."""
    N, cp = 64, 16
    L = N + cp
    rng = np.random.default_rng(2)
    for every, n_data in ((70000, 70010), (3, 90001)):
        tx = om.TxEngine(N, cp, N - 2, 60, (1, every), "QPSK")
        data = (rng.standard_normal((n_data, L)) + 1j * rng.standard_normal((n_data, L))).astype(np.complex64)
        d_in = om.DeviceBuffer(data.nbytes).upload(data)
        n_out = tx.mux_symbols(n_data)
        d_out = om.DeviceBuffer(n_out * L * 8)
        assert tx.mux(d_in, n_data, d_out) == n_out
        out = d_out.download(np.complex64, n_out * L).reshape(n_out, L)
        sync = tx.sync_symbol()[0]
        is_sync = np.arange(n_out) % (1 + every) == 0
        assert is_sync.sum() == n_out - n_data
        assert np.array_equal(out[is_sync], np.broadcast_to(sync, (int(is_sync.sum()), L)))
        assert np.array_equal(out[~is_sync], data)


def test_random_bit_source_matches_its_definition(om):
    tx = om.TxEngine(64, 16, 62, 60)
    for seed, off, n in [(0, 0, 128), (20260101, 0, 5000), (20260101, 4097, 777), ((7 << 32) | 9, (1 << 33) + 5, 1000), (1, 127, 2)]:
        d = om.DeviceBuffer(max(n, 8))
        tx.random_bits(seed, off, d, n)
        assert np.array_equal(d.download(np.uint8, n), orc.random_bits(seed, off, n)), (seed, off, n)
    d = om.DeviceBuffer(128)
    tx.random_bits(0, 0, d, 128)                                   # Philox known answer: counter 0, key 0
    w = np.packbits(d.download(np.uint8, 128).reshape(4, 32)[:, ::-1], axis=1).view(">u4").ravel()
    assert [int(x) for x in w] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


def test_blocks_chained_like_the_flowgraph(golden):
    """The six `txOFDM` blocks, constructed and wired as RXtransmit_6.grc does, called the way the reference's offline harness
    calls blocks (work() by hand): with pilot_locations = [] they replay the reference's bit fixture into its IQ fixture."""
    import txOFDM
    fx = golden("ref_fixtures.npz")
    bits = fx["tx_bits"][0].astype(np.uint8)
    cm, omod = txOFDM.ConstellationModulation("QPSK"), txOFDM.OFDM_Modulation(64, [], num_data_bins=60)
    ifft, cpb, mux = txOFDM.IFFT(64), txOFDM.CyclicPrefix(64, 16), txOFDM.SynchDataMux(64, 16, 23, 3, 62)

    def run(blk, x, n_out):
        out = np.zeros(n_out, np.complex64)
        n = (getattr(blk, "general_work", None) or blk.work)([x], [out])
        return out[:n]
    x = run(cm, bits, len(bits) // 2)
    x = run(omod, x, 180 * 64 + 13)                                 # scheduler slack: only whole symbols move
    assert len(x) == 180 * 64
    x = run(ifft, x, 180 * 64)
    x = run(cpb, x, 180 * 80)
    x = run(mux, x, 240 * 80 + 79)
    assert len(x) == 19200 and relerr(x, fx["tx_online"][0]) < TOL
    # stream in two uneven pieces: same output (the blocks keep no symbol state between calls)
    a = run(txOFDM.IFFT(64), np.concatenate([run(omod, run(cm, bits[:60 * 2 * 7], 60 * 7), 7 * 64)]), 7 * 64)
    assert np.array_equal(a, run(ifft, run(omod, run(cm, bits, len(bits) // 2), 180 * 64), 180 * 64)[:7 * 64])
    src = txOFDM.random_bit_source(seed=42)
    b1, b2 = np.zeros(1000, np.uint8), np.zeros(500, np.uint8)
    assert src.work([], [b1]) == 1000 and src.work([], [b2]) == 500
    assert np.array_equal(np.concatenate([b1, b2]), orc.random_bits(42, 0, 1500))


@pytest.mark.parametrize("N,cp,Kd,mod", [(64, 15, 60, "16QAM"), (256, 19, 180, "64QAM"), (2048, 143, 1200, "QPSK"),
                                          (2048, 144, 1200, "64QAM"), (1024, 72, 600, "16QAM")])
def test_fused_modulator_layout_fallbacks_equal_the_oracle(om, N, cp, Kd, mod):
    """The fused kernel picks its fast paths from the layout it is handed: wide bit loads need the stream 4-byte aligned, the
    16-byte stores an even cp and symbol length and an aligned output.  An odd cp, an output that starts 8 bytes into a 16-byte
    line, an odd frame stride and a bit stream at an odd address must give the same samples as the aligned call and the oracle."""
    import torch
    n_sym, n_frames = 8, 3
    L = N + cp
    rng = np.random.default_rng(N + cp)
    tx = om.TxEngine(N, cp, N - 2, Kd, (1, 3), mod)
    nb = tx.bits_per_frame(n_sym)
    bits = rng.integers(0, 2, (n_frames, nb)).astype(np.uint8)
    ref = np.stack([orc.tx_modulate(bits[f], N, cp, N - 2, Kd, n_sym, modulation=mod) for f in range(n_frames)])
    dev = torch.device("cuda", 0)
    # aligned call
    d_bits = torch.from_numpy(bits).to(dev)
    d_iq = torch.zeros((n_frames, n_sym * L, 2), dtype=torch.float32, device=dev)
    tx.modulate_frames(d_bits.data_ptr(), n_frames, n_sym, d_iq.data_ptr())
    torch.cuda.synchronize()
    y0 = d_iq.cpu().numpy().view(np.complex64).reshape(n_frames, n_sym * L)
    assert relerr(y0, ref) < TOL
    # bit stream at an odd address (frames back to back: the stride nb keeps every frame misaligned too when nb is even)
    raw = torch.zeros(n_frames * nb + 8, dtype=torch.uint8, device=dev)
    raw[1:1 + n_frames * nb] = d_bits.reshape(-1)
    # output 8 bytes into a 16-byte line, frames one sample further apart than they need to be (odd stride)
    stride = n_sym * L + 1
    out = torch.zeros((n_frames * stride + 4, 2), dtype=torch.float32, device=dev)
    tx.modulate_frames(raw.data_ptr() + 1, n_frames, n_sym, out.data_ptr() + 8, frame_stride=stride)
    torch.cuda.synchronize()
    o = out.cpu().numpy().view(np.complex64).reshape(-1)
    y1 = np.stack([o[1 + f * stride:1 + f * stride + n_sym * L] for f in range(n_frames)])
    assert np.array_equal(y1, y0)
    assert o[0] == 0 and all(o[1 + f * stride + n_sym * L] == 0 for f in range(n_frames))     # nothing written outside the frames


@pytest.mark.parametrize("N,cp,Kd,S,D,n_sym", [(1024, 72, 600, 2, 3, 15), (64, 16, 60, 2, 3, 10), (2048, 144, 1200, 1, 2, 9), (4096, 288, 2400, 3, 1, 8)])
def test_fused_modulator_with_other_sync_patterns(om, N, cp, Kd, S, D, n_sym):
    """[S, D] patterns other than [1, 3], incl. S > 1 (every sync symbol carries ZC segment 0: the reference never advances
    `synch_state`, MultiAntennaSystem.py:143-147).  From 1024-pt up the kernel copies sync symbols from the handle's finished
    ones, row r of [S][L] for the r-th sync symbol of a pattern, and its grid stride is kept coprime with S + D."""
    n_frames = 3
    rng = np.random.default_rng(N + S)
    tx = om.TxEngine(N, cp, N - 2, Kd, (S, D), "QPSK")
    nb = tx.bits_per_frame(n_sym)
    bits = rng.integers(0, 2, (n_frames, nb)).astype(np.uint8)
    ref = np.stack([orc.tx_modulate(bits[f], N, cp, N - 2, Kd, n_sym, synch_dat=(S, D)) for f in range(n_frames)])
    d_bits = om.DeviceBuffer(bits.nbytes).upload(bits)
    d_iq = om.DeviceBuffer(ref.size * 8)
    tx.modulate_frames(d_bits, n_frames, n_sym, d_iq)
    y = d_iq.download(np.complex64, ref.size).reshape(ref.shape)
    assert relerr(y, ref) < TOL
    L = N + cp
    sync = tx.sync_symbol()
    for s in range(n_sym):
        if s % (S + D) < S:
            assert np.array_equal(y[0, s * L:(s + 1) * L], sync[s % (S + D)])
