"""The decomposed live transmitter: GNU Radio module `txOFDM`.

    random_bit_source() -> ConstellationModulation(modulation) -> OFDM_Modulation(fft_size, pilot_locations)
      -> IFFT(fft_size) -> CyclicPrefix(fft_size, cp_size) -> SynchDataMux(fft_size, cp_size, prime_no, synch_every, synch_length)

The reference only NAMES these blocks, with exactly these parameters, in a flowgraph
(LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc:701-975, wired :1819-1854); the `txOFDM` module itself is absent from the repository.
What each stage has to compute is fixed by the modulator the repository does hold -- MultiAntennaSystem.multi_ant_binary_map /
multi_ant_symb_gen (LEGACY/gr-ofdm-rx/python/txrx_mod/MultiAntennaSystem.py:150-218) and SynchSignal (SynchSignal.py:13-30):
chained with pilot_locations = [], the six blocks emit the same IQ stream as that modulator (tests/test_gpu_tx_stages.py:
equal to the fused HIP kernel bit for bit, equal to the reference's `tx_data_online` fixture at 1e-5).  Pilots, and every
parameter the flowgraph leaves out (number of data bins), are extensions with keyword defaults: parity unpinned.

All ports are plain item streams (bytes for bits, complex64 otherwise).  Stages whose input and output group sizes differ are
`gr.basic_block`s that move whole OFDM symbols per call (`forecast` / `general_work`); every sample is computed by the HIP
kernels behind `TxEngine` (ofdm_tx_random_bits / _map / _grid / _ifft_cp / _mux of include/ofdm_mi355x.h).
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib
from .engine import DeviceBuffer, TxEngine
from .gr_compat import basic_block, decim_block, sync_block


def _device() -> int:
    return int(os.environ.get("OFDM_MI355X_DEVICE", "0"))


class _Scratch:
    """Two device buffers that grow on demand (stream blocks are called with a few thousand items at a time)."""

    def __init__(self, device):
        self.device = device
        self.bufs = [None, None]

    def get(self, which, nbytes):
        b = self.bufs[which]
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = self.bufs[which] = DeviceBuffer(max(int(nbytes) * 2, 4096), self.device)
        return b


def _default_data_bins(fft_size, n_pilots):
    """Occupied span when the flowgraph does not say: every bin but DC and Nyquist, like synch_length = fft_size - 2."""
    return int(fft_size) - 2 - int(n_pilots)


class random_bit_source(sync_block):  # noqa: N801  (the flowgraph's block name)
    """txOFDM.random_bit_source() -- source of uniform random bits, one per output byte.  The stream is a pure function of
    (seed, item index) (counter-based Philox on the device), so any window of it can be regenerated."""

    def __init__(self, seed=20260101):
        sync_block.__init__(self, name="random_bit_source", in_sig=None, out_sig=[np.uint8])
        self.seed = int(seed)
        self.offset = 0
        self._eng = TxEngine(64, 16, 62, 60, device=_device())
        self._scratch = _Scratch(_device())

    def work(self, input_items, output_items):
        out = output_items[0]
        n = len(out)
        buf = self._scratch.get(0, n)
        self._eng.random_bits(self.seed, self.offset, buf, n)
        out[:] = buf.download(np.uint8, n)
        self.offset += n
        return n


class ConstellationModulation(decim_block):
    """txOFDM.ConstellationModulation(modulation) -- bits (one per byte, MSB of a symbol first) -> constellation points
    (MultiAntennaSystem.py:150-178 for BPSK / QPSK; TS 36.211 Gray maps for "16QAM" / "64QAM")."""

    def __init__(self, modulation="QPSK"):
        self.modulation = modulation
        self.bits_per_symbol = _lib.MODULATION_BITS[str(modulation).upper().replace("-", "")]
        decim_block.__init__(self, name="ConstellationModulation", in_sig=[np.uint8], out_sig=[np.complex64],
                             decim=self.bits_per_symbol)
        self._eng = TxEngine(64, 16, 62, 60, modulation=modulation, device=_device())
        self._scratch = _Scratch(_device())

    def work(self, input_items, output_items):
        out = output_items[0]
        n = len(out)
        bits = np.ascontiguousarray(input_items[0][:n * self.bits_per_symbol], dtype=np.uint8)
        if bits.size < n * self.bits_per_symbol:
            raise ValueError("ConstellationModulation: %d output items need %d bits, got %d" % (n, n * self.bits_per_symbol, bits.size))
        d_in = self._scratch.get(0, bits.nbytes).upload(bits)
        d_out = self._scratch.get(1, n * 8)
        self._eng.map(d_in, n, d_out, _lib.BITS_UNPACKED)
        out[:] = d_out.download(np.complex64, n)
        return n


class _SymbolRateBlock(basic_block):
    """A stage that turns whole groups of `in_group` items into groups of `out_group` items."""

    def _init_rates(self, name, in_group, out_group):
        basic_block.__init__(self, name=name, in_sig=[np.complex64], out_sig=[np.complex64])
        self.in_group, self.out_group = int(in_group), int(out_group)
        self.set_output_multiple(self.out_group)
        self.set_relative_rate(self.out_group / self.in_group)
        self._scratch = _Scratch(_device())

    def forecast(self, noutput_items, ninputs=1):
        need = (int(noutput_items) // self.out_group) * self.in_group
        return [need] * (ninputs if isinstance(ninputs, int) else len(ninputs))

    def general_work(self, input_items, output_items):
        in0, out = input_items[0], output_items[0]
        k = min(len(in0) // self.in_group, len(out) // self.out_group)
        if k > 0:
            src = np.ascontiguousarray(in0[:k * self.in_group], dtype=np.complex64)
            d_in = self._scratch.get(0, src.nbytes).upload(src)
            d_out = self._scratch.get(1, k * self.out_group * 8)
            self._run(d_in, k, d_out)
            out[:k * self.out_group] = d_out.download(np.complex64, k * self.out_group)
        self.consume_each(k * self.in_group)
        return k * self.out_group

    # the offline harness style of the reference (topblock.py:84-88) calls work(); same thing here
    work = general_work


class OFDM_Modulation(_SymbolRateBlock):  # noqa: N801
    """txOFDM.OFDM_Modulation(fft_size, pilot_locations) -- constellation points -> resource-grid rows: `num_data_bins` symbols
    per OFDM symbol on the occupied bins [-K/2..-1, 1..K/2] (K = num_data_bins + len(pilot_locations)) in that order
    (MultiAntennaSystem.py:135-139,182-183), `pilot_value` on the signed offsets `pilot_locations`, zero elsewhere (DC too)."""

    def __init__(self, fft_size, pilot_locations=(), num_data_bins=None, pilot_value=1.0 + 0.0j):
        self.fft_size = int(fft_size)
        self.pilot_locations = [int(p) for p in pilot_locations]
        self.num_data_bins = _default_data_bins(fft_size, len(self.pilot_locations)) if num_data_bins is None else int(num_data_bins)
        self.pilot_value = complex(pilot_value)
        self._eng = TxEngine(self.fft_size, 0, self.fft_size - 2, self.num_data_bins, device=_device())
        self._eng.set_pilots(self.pilot_locations, self.pilot_value)
        self._init_rates("OFDM_Modulation", self.num_data_bins, self.fft_size)

    def _run(self, d_in, k, d_out):
        self._eng.grid(d_in, k, d_out)


class IFFT(_SymbolRateBlock):
    """txOFDM.IFFT(fft_size) -- one inverse FFT per `fft_size` items, numpy.fft.ifft scaling (MultiAntennaSystem.py:199)."""

    def __init__(self, fft_size):
        self.fft_size = int(fft_size)
        self._eng = TxEngine(self.fft_size, 0, self.fft_size - 2, self.fft_size - 2, device=_device())
        self._init_rates("IFFT", self.fft_size, self.fft_size)

    def _run(self, d_in, k, d_out):
        self._eng.ifft_cp(d_in, k, d_out, do_ifft=True, add_cp=False)


class CyclicPrefix(_SymbolRateBlock):
    """txOFDM.CyclicPrefix(fft_size, cp_size) -- prepend the last `cp_size` samples of every symbol, then normalise the
    CP-extended symbol to unit power exactly as the modulator does (scale by sqrt(L/E), divide by the standard deviation;
    MultiAntennaSystem.py:200-218)."""

    def __init__(self, fft_size, cp_size):
        self.fft_size, self.cp_size = int(fft_size), int(cp_size)
        self._eng = TxEngine(self.fft_size, self.cp_size, self.fft_size - 2, self.fft_size - 2, device=_device())
        self._init_rates("CyclicPrefix", self.fft_size, self.fft_size + self.cp_size)

    def _run(self, d_in, k, d_out):
        self._eng.ifft_cp(d_in, k, d_out, do_ifft=False, add_cp=True)


class SynchDataMux(_SymbolRateBlock):
    """txOFDM.SynchDataMux(fft_size, cp_size, prime_no, synch_every, synch_length) -- one Zadoff-Chu sync symbol (root
    `prime_no` on `synch_length` bins, SynchSignal.py:13-30; synthesised once through the same IFFT + CP + normalisation) in
    front of every `synch_every` data symbols: the [1, synch_every] pattern the receivers are configured with."""

    def __init__(self, fft_size, cp_size, prime_no, synch_every, synch_length):
        self.fft_size, self.cp_size = int(fft_size), int(cp_size)
        self.prime_no, self.synch_every, self.synch_length = int(prime_no), int(synch_every), int(synch_length)
        self._eng = TxEngine(self.fft_size, self.cp_size, self.synch_length, self.fft_size - 2, synch_dat=(1, self.synch_every),
                             zc_root=self.prime_no, device=_device())
        L = self.fft_size + self.cp_size
        self._init_rates("SynchDataMux", self.synch_every * L, (self.synch_every + 1) * L)

    @property
    def synch_symbol(self):
        return self._eng.sync_symbol()[0]

    def _run(self, d_in, k, d_out):
        self._eng.mux(d_in, k * self.synch_every, d_out)
