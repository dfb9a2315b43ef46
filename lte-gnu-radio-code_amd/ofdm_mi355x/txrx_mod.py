"""Mirror of the reference's offline modulator classes (LEGACY/gr-ofdm-rx/python/txrx_mod/), single antenna.

    OFDM(len_CP, num_used_bins, modulation_type, NFFT, delta_f)                       OFDM.py:4-15
    SynchSignal(len_CP, num_synch_bins, num_ant_txrx, NFFT, synch_data)               SynchSignal.py:4-38
    MultiAntennaSystem(OFDM_data, num_ant_txrx, MIMO_method, all_bins, num_symbols, symbol_pattern, fs,
                       channel_profile, diagnostic, wireless_channel, stream_size, data_only_bins, ref_only_bins)
        .multi_ant_binary_map(Caz, binary_info, synch_data)                            MultiAntennaSystem.py:113-187
        .multi_ant_symb_gen(num_symbols)        -> buffer_data_tx_time                 :189-218
        .rx_signal_gen()                         -> buffer_data_rx_time                 :221-231
        .additive_noise(SNR_type, SNR_dB, wireless_channel, sig_datatype)               :235-260

Same constructor orders, method names and result buffers, so the reference's `SDRScript.py:112-146` flow runs
unchanged; the bit map, grid fill, IFFT + CP + normalisation, channel convolution and AWGN all execute in the HIP
kernels (TxEngine).  Differences: one antenna only (the reference's 2-antenna branches print "not implemented" and
exit, MultiAntennaSystem.py:184-186); `buffer_data_tx` (the frequency-domain grid) is not materialised because map,
grid and IFFT are one fused kernel; the noise is Philox/Box-Muller on the device, reproducible from `noise_seed`.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .engine import DeviceBuffer, TxEngine, zadoff_chu

REF_TAPS = np.array([0.3977, 0.7954 - 0.3977j, -0.1988, 0.0994, -0.0398])     # MultiAntennaSystem.py:64


class OFDM:
    def __init__(self, len_CP, num_used_bins, modulation_type, NFFT, delta_f):
        self.len_CP = len_CP
        self.num_used_bins = num_used_bins
        self.modulation_type = modulation_type
        self.NFFT = NFFT
        self.bin_spacing = delta_f
        self.num_bits_bin = _lib.MODULATION_BITS[str(modulation_type).upper().replace("-", "")]


class SynchSignal:
    def __init__(self, len_CP, num_synch_bins, num_ant_txrx, NFFT, synch_data):
        self.synch_state = 0
        self.len_CP = len_CP
        self.num_used_bins = int(num_synch_bins)
        self.num_ant = num_ant_txrx
        self.NFFT = int(NFFT)
        h = int(self.num_used_bins / 2)
        self.used_bins0 = list(range(-h, 0)) + list(range(1, h + 1))
        self.used_bins = (self.NFFT + np.array(self.used_bins0)) % self.NFFT
        self.synch_data = synch_data
        self.M = np.array([synch_data[0], self.num_used_bins])
        self.MM = int(np.prod(self.M))
        self.prime = 23
        self.ZChu0 = zadoff_chu(self.MM, self.prime)


class MultiAntennaSystem:
    def __init__(self, OFDM_data, num_ant_txrx, MIMO_method, all_bins, num_symbols, symbol_pattern, fs, channel_profile,
                 diagnostic, wireless_channel, stream_size, data_only_bins, ref_only_bins, device=0, noise_seed=0):
        if num_ant_txrx != 1:
            raise NotImplementedError("single antenna only (the reference's multi-antenna branches exit unimplemented)")
        self.OFDM_data = OFDM_data
        self.NFFT = int(OFDM_data.NFFT)
        self.len_CP = int(OFDM_data.len_CP)
        self.num_ant_txrx = 1
        self.all_bins = np.asarray(all_bins)
        self.used_bins = (self.NFFT + self.all_bins) % self.NFFT
        self.num_symbols = int(num_symbols)
        self.symbol_pattern = np.asarray(symbol_pattern)
        self.fs = fs
        self.wireless_channel = wireless_channel
        self.stream_size = stream_size
        self.num_used_bins = OFDM_data.num_used_bins
        self.max_impulse = self.NFFT
        self.device = device
        self.noise_seed = noise_seed
        self.symb_len_total = (self.NFFT + self.len_CP) * self.num_symbols
        self.channel_time = np.zeros((1, 1, self.max_impulse), dtype=complex)
        if wireless_channel == "AWGN":
            self.channel_time[0, 0, 1] = 1                                    # :81-82 (a one-sample delay)
            self._n_taps = 2
        else:
            self.channel_time[0, 0, 0:len(REF_TAPS)] = REF_TAPS / np.linalg.norm(REF_TAPS)   # :86
            self._n_taps = len(REF_TAPS)
        self.genie_chan_time = self.channel_time
        self.buffer_data_tx_time = np.zeros((1, self.symb_len_total), dtype=complex)
        self.buffer_data_rx_time = np.zeros((1, self.symb_len_total + self.max_impulse - 1), dtype=complex)
        self._bits = None
        self._engine = None

    def multi_ant_binary_map(self, Caz, binary_info, synch_data):
        pat = self.symbol_pattern.astype(int)
        S = int(synch_data[0])
        D = int(synch_data[1])
        expect = np.tile(np.concatenate((np.zeros(S, int), np.ones(D, int))), len(pat) // (S + D) + 1)[:len(pat)]
        if not np.array_equal(pat, expect):
            raise ValueError("symbol_pattern must repeat [0]*S + [1]*D (what SDRScript.py:78-79 builds)")
        self.zchu = Caz.ZChu0
        self.synch_data = synch_data
        self._engine = TxEngine(self.NFFT, self.len_CP, Caz.num_used_bins, self.num_used_bins, (S, D),
                                self.OFDM_data.modulation_type, Caz.prime, device=self.device)
        need = self._engine.bits_per_frame(self.num_symbols)
        bits = np.ascontiguousarray(np.asarray(binary_info)[0, :need], dtype=np.uint8)
        if bits.size < need:
            raise IndexError("binary_info holds %d bits, %d needed" % (bits.size, need))
        self._bits = bits

    def multi_ant_symb_gen(self, num_symbols):
        if self._engine is None:
            raise RuntimeError("call multi_ant_binary_map first")
        n = int(num_symbols)
        L = self.NFFT + self.len_CP
        d_bits = DeviceBuffer(max(8, self._bits.nbytes), self.device).upload(self._bits)
        self._d_tx = DeviceBuffer(n * L * 8, self.device)
        self._engine.modulate_frames(d_bits, 1, n, self._d_tx, n * L, _lib.BITS_UNPACKED)
        self.buffer_data_tx_time[0, :n * L] = self._d_tx.download(np.complex64, n * L)

    def _channel(self, noise_var):
        n_in = self.symb_len_total
        n_out = n_in + self.max_impulse - 1
        taps = self.channel_time[0, 0, :self._n_taps].astype(np.complex64)
        d_t = DeviceBuffer(taps.nbytes, self.device).upload(taps)
        d_y = DeviceBuffer(n_out * 8, self.device)
        d_y.upload(np.zeros(n_out, np.complex64))
        n_conv = n_in + self._n_taps - 1                                       # the rest of the N-tap tail is exactly zero
        self._engine.channel(self._d_tx, 1, n_in, n_in, d_t, self._n_taps, d_y, n_out, n_conv, noise_var=noise_var,
                             seed=self.noise_seed)
        return d_y.download(np.complex64, n_out)

    def rx_signal_gen(self):
        self.buffer_data_rx_time[0, :] = self._channel(0.0)

    def additive_noise(self, SNR_type, SNR_dB, wireless_channel, sig_datatype):
        self.SNR_lin = 10 ** (SNR_dB / 10)
        sig_pow = np.var(self.buffer_data_tx_time)                             # :237
        bits_per_symb = self.num_used_bins * self.OFDM_data.num_bits_bin
        samp_per_symb = self.NFFT + self.len_CP
        if SNR_type == "Digital":
            self.noise_var = (1 / bits_per_symb) * samp_per_symb * sig_pow * 10 ** (-SNR_dB / 10)   # :244
        else:
            self.noise_var = sig_pow * 10 ** (-SNR_dB / 10)                    # :246
        self.SNR_analog = sig_pow / self.noise_var
        if sig_datatype != "Complex":
            raise NotImplementedError("complex baseband only")
        y = self._channel(float(self.noise_var))
        n_conv = self.symb_len_total + self._n_taps - 1
        # the reference adds noise over the whole buffer incl. the all-zero tail (:258-260); the device adds it to the
        # n_conv samples it convolves, the remaining tail samples stay zero
        self.buffer_data_rx_time[0, :] = y
        self._noise_len = n_conv
