"""Host logic of the `txOFDM` blocks (no GPU): item-rate bookkeeping (forecast / consume / produce in whole OFDM symbols),
constructor signatures as the reference's flowgraph calls them, and the wiring of each block to its C-ABI stage.  The device
engine is replaced by a stand-in that computes the stages with the oracle, so the six blocks chained here must reproduce
`oracle.tx_modulate` -- on the GPU the same chain is checked against the HIP kernels (tests/test_gpu_tx_stages.py)."""
import numpy as np
import pytest

from oracle import ofdm_oracle as orc
from ofdm_mi355x import tx_blocks


class _Buf:
    def __init__(self, nbytes, device=0):
        self.nbytes = nbytes
        self.arr = None

    def upload(self, a):
        self.arr = np.array(a)
        return self

    def download(self, dtype, count):
        return np.asarray(self.arr).ravel()[:count].astype(dtype)

    def free(self):
        pass


class _Eng:
    MOD = {1: "BPSK", 2: "QPSK", 4: "16QAM", 6: "64QAM"}

    def __init__(self, nfft, cp_len, num_synch_bins, num_data_bins, synch_dat=(1, 3), modulation="QPSK", zc_root=23, device=0):
        self.N, self.cp, self.Ks, self.Kd, self.sd, self.mod, self.root = nfft, cp_len, num_synch_bins, num_data_bins, synch_dat, modulation, zc_root
        self.pil, self.pv = [], 1.0

    def random_bits(self, seed, offset, d, n):
        d.arr = orc.random_bits(seed, offset, n)

    def map(self, d_in, n, d_out, mode):
        d_out.arr = orc.map_bits(d_in.arr[:n * orc.BITS_PER_SYMBOL[self.mod]], self.mod)

    def set_pilots(self, loc, val):
        self.pil, self.pv = list(loc), val

    def grid(self, d_in, k, d_out):
        d_out.arr = orc.tx_stage_grid(d_in.arr[:k * self.Kd], self.N, self.Kd, self.pil, self.pv)

    def ifft_cp(self, d_in, k, d_out, do_ifft=True, add_cp=True):
        x = np.asarray(d_in.arr)[:k * self.N].reshape(k, self.N)
        if do_ifft:
            x = orc.tx_stage_ifft(x)
        d_out.arr = orc.tx_stage_cp(x, self.cp) if add_cp else x

    def mux(self, d_in, n, d_out):
        L = self.N + self.cp
        d_out.arr = orc.tx_stage_mux(np.asarray(d_in.arr)[:n * L].reshape(n, L), self.N, self.cp, self.root, self.sd[1], self.Ks)

    def sync_symbol(self):
        return orc.tx_stage_mux(np.zeros((1, self.N + self.cp)), self.N, self.cp, self.root, self.sd[1], self.Ks)[:1]


@pytest.fixture()
def blocks(monkeypatch):
    monkeypatch.setattr(tx_blocks, "TxEngine", _Eng)
    monkeypatch.setattr(tx_blocks, "DeviceBuffer", _Buf)
    return tx_blocks


def _run(blk, x, n_out, dtype=np.complex64):
    out = np.zeros(n_out, dtype)
    fn = getattr(blk, "general_work", None) or blk.work
    n = fn([x], [out])
    return out[:n], n


def test_chain_as_the_flowgraph_wires_it(blocks):
    """RXtransmit_6.grc:1819-1854 with fft1 = 64: every block constructed with the flowgraph's own arguments."""
    fft1 = 64
    src = blocks.random_bit_source()
    cm = blocks.ConstellationModulation("QPSK")
    om = blocks.OFDM_Modulation(fft1, [-21, -7, 7, 21])
    ifft = blocks.IFFT(fft1)
    cpb = blocks.CyclicPrefix(fft1, fft1 // 4)
    mux = blocks.SynchDataMux(fft1, fft1 // 4, 47, 3, fft1 - 2)
    assert om.num_data_bins == 58 and cm.decimation() == 2
    n_ofdm = 6
    bits, _ = _run(src, None, n_ofdm * 58 * 2, np.uint8)
    assert src.offset == len(bits) and np.array_equal(bits, orc.random_bits(src.seed, 0, len(bits)))
    sym, _ = _run(cm, bits, n_ofdm * 58)
    grid, _ = _run(om, sym, n_ofdm * 64)
    assert om.consumed == [n_ofdm * 58]
    t, _ = _run(ifft, grid, n_ofdm * 64)
    c, _ = _run(cpb, t, n_ofdm * 80)
    out, n = _run(mux, c, 8 * 80)
    assert n == 8 * 80 and mux.consumed == [6 * 80]
    ref = orc.tx_stage_mux(orc.tx_stage_cp(orc.tx_stage_ifft(orc.tx_stage_grid(orc.map_bits(bits, "QPSK"), 64, 58, [-21, -7, 7, 21])), 16),
                           64, 16, 47, 3, 62).ravel()
    assert np.max(np.abs(out - ref)) < 1e-6


def test_chain_without_pilots_is_the_reference_modulator(blocks, golden):
    fx = golden("ref_fixtures.npz")
    bits = fx["tx_bits"][0].astype(np.uint8)
    cm, om = blocks.ConstellationModulation("QPSK"), blocks.OFDM_Modulation(64, [], num_data_bins=60)
    ifft, cpb, mux = blocks.IFFT(64), blocks.CyclicPrefix(64, 16), blocks.SynchDataMux(64, 16, 23, 3, 62)
    x, _ = _run(cm, bits, len(bits) // 2)
    x, _ = _run(om, x, 180 * 64)
    x, _ = _run(ifft, x, 180 * 64)
    x, _ = _run(cpb, x, 180 * 80)
    x, n = _run(mux, x, 240 * 80)
    assert n == 19200 and np.max(np.abs(x - fx["tx_online"][0])) / np.max(np.abs(fx["tx_online"][0])) < 1e-6


def test_whole_symbols_only(blocks):
    """A scheduler buffer that is not a whole number of symbols: only whole groups are consumed and produced."""
    om = blocks.OFDM_Modulation(64, [], num_data_bins=60)
    x = (np.arange(60 * 3 + 17) + 1).astype(np.complex64)
    out, n = _run(om, x, 64 * 2 + 5)                       # room for two rows only
    assert n == 128 and om.consumed == [120]
    assert om.forecast(128, 1) == [120] and om.forecast(127, 1) == [60] and om.forecast(63, 1) == [0]
    assert om.output_multiple() == 64 and abs(om.relative_rate() - 64 / 60) < 1e-12
    mux = blocks.SynchDataMux(64, 16, 23, 3, 62)
    assert mux.forecast(4 * 80, 1) == [3 * 80] and mux.output_multiple() == 320
    out, n = _run(mux, np.ones(3 * 80 + 79, np.complex64), 8 * 80 - 1)
    assert n == 320 and mux.consumed == [240]
    out, n = _run(mux, np.ones(100, np.complex64), 320)
    assert n == 0 and mux.consumed == [0]


def test_constellation_block_rejects_short_input(blocks):
    cm = blocks.ConstellationModulation("16QAM")
    with pytest.raises(ValueError):
        cm.work([np.zeros(7, np.uint8)], [np.zeros(2, np.complex64)])
