"""Host-side engines over the C ABI: receive chain, transmit chain, loop-back channel, de-mapper.

Nothing here computes samples on the CPU; NumPy is used for buffers and for the small fp64
constant tables the reference exposes as block attributes (Zadoff-Chu sequence, bin lists).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import BITS_NONE, BITS_PACKED, BITS_UNPACKED, MODULATION_BITS, check, ptr


def bins_p(num_bins: int, nfft: int) -> np.ndarray:
    """binsP(K): [-K/2..-1, 1..K/2] + N mod N  (gr-utsa_ofdm/python/SynchAndChanEst.py:38-41,66-70)."""
    h = int(num_bins / 2)
    return (np.array(list(range(-h, 0)) + list(range(1, h + 1)), dtype=np.int64) + nfft) % nfft


def zadoff_chu(mm: int, root: int, parity_of: int | None = None) -> np.ndarray:
    """SynchAndChanEst.py:52-59 (root 23) / gr-RXOFDM synch_and_chan_est.py:54-64 (root 37)."""
    par = mm if parity_of is None else parity_of
    x0 = np.arange(mm, dtype=np.float64)
    q = x0 ** 2 / 2 if par % 2 == 0 else x0 * (x0 + 1) / 2
    return np.exp(-1j * (2 * np.pi / mm) * root * q)


def _mod_bits(modulation) -> int:
    if isinstance(modulation, str):
        return MODULATION_BITS[modulation.upper().replace("-", "")]
    return int(modulation)


class DeviceBuffer:
    """Plain HBM allocation through the C ABI (for hosts that do not use torch)."""

    def __init__(self, nbytes: int, device: int = 0):
        self.lib = _lib.load()
        self.device = device
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(self.lib.ofdm_device_malloc(device, C.byref(p), self.nbytes))
        self._ptr = p.value or 0

    def data_ptr(self) -> int:
        return self._ptr

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(self.lib.ofdm_memcpy_h2d(self.device, ptr(self), ptr(arr), arr.nbytes))
        return self

    def download(self, dtype, count: int) -> np.ndarray:
        """Blocking device-to-host copy; waits for ALL streams of the device first (the batch entry points
        are asynchronous on the handles' non-blocking streams)."""
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self.lib.ofdm_device_synchronize(self.device))
        check(self.lib.ofdm_memcpy_d2h(self.device, ptr(out), ptr(self), out.nbytes))
        return out

    def free(self):
        if self._ptr:
            self.lib.ofdm_device_free(self.device, C.c_void_p(self._ptr))
            self._ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def count_bit_errors(d_a, d_b, n_bytes: int, d_count, stream=None, device: int = 0):
    """*d_count (device uint64, zeroed by the caller) += popcount(a ^ b) over n_bytes of two device byte strings."""
    check(_lib.load().ofdm_count_bit_errors(int(device), ptr(d_a), ptr(d_b), int(n_bytes), ptr(d_count), ptr(stream)))


class RxEngine:
    """Receive chain handle (sync search, LS channel estimate, FFT + equalise, de-map)."""

    def __init__(self, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr,
                 scale_factor_gate=0.7, compat=_lib.COMPAT_UTSA, modulation="QPSK", device=0):
        self.lib = _lib.load()
        self.cfg = _lib.RxCfg(int(num_ofdm_symb), int(nfft), int(cp_len), int(num_synch_bins), int(synch_dat[0]),
                              int(synch_dat[1]), int(num_data_bins), float(snr), float(scale_factor_gate),
                              int(compat), _mod_bits(modulation), int(device), 0)
        h = C.c_void_p()
        check(self.lib.ofdm_rx_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        self.report = _lib.RxReport()

    # ---- stream block -----------------------------------------------------------------------
    def work(self, in0: np.ndarray, out: np.ndarray) -> int:
        """Returns the raw status; the caller copies `self.report` (valid even when the reference would have
        raised after its sync search) and then passes the status to `_lib.check`."""
        in0 = np.ascontiguousarray(in0, dtype=np.complex64)
        if out.dtype != np.complex64 or not out.flags.c_contiguous:
            tmp = np.ascontiguousarray(out, dtype=np.complex64)
            rc = self.lib.ofdm_rx_work(self._h, ptr(in0), in0.size, ptr(tmp), tmp.size, C.byref(self.report))
            if rc >= 0:
                out[...] = tmp
            return int(rc)
        return int(self.lib.ofdm_rx_work(self._h, ptr(in0), in0.size, ptr(out), out.size, C.byref(self.report)))

    def state(self, row: int = 0):
        c = self.cfg
        mm = c.synch_S * c.num_synch_bins
        H = np.zeros(c.nfft, np.complex64)
        ht = np.zeros(c.nfft, np.complex64)
        esf = np.zeros(mm, np.complex64)
        eqg = np.zeros(c.num_synch_bins, np.complex64)
        edf = np.zeros((c.num_ofdm_symb, c.num_data_bins), np.complex64)
        check(self.lib.ofdm_rx_get_state(self._h, row, ptr(H), ptr(ht), ptr(esf), ptr(eqg), ptr(edf)))
        return dict(chan_freq=H, chan_time=ht, synch_freq=esf, eq_gain=eqg, data_freq=edf)

    # ---- frame batches on device buffers ----------------------------------------------------
    def data_symbols_per_frame(self, frame_len: int) -> int:
        c = self.cfg
        return (frame_len // (c.nfft + c.cp_len)) // (c.synch_S + c.synch_D) * c.synch_D

    def reserve(self, n_frames: int):
        check(self.lib.ofdm_rx_reserve(self._h, int(n_frames)))

    def demod_frames(self, d_iq, n_frames, frame_stride, frame_len, d_eq=None, d_bits=None,
                     bits_mode=BITS_NONE, d_tsr=None, stream=None) -> int:
        return int(check(self.lib.ofdm_rx_demod_frames(self._h, ptr(d_iq), int(n_frames), int(frame_stride),
                                                       int(frame_len), ptr(d_eq), ptr(d_bits), int(bits_mode),
                                                       ptr(d_tsr), ptr(stream))))

    def set_profiling(self, enable: bool = True):
        check(self.lib.ofdm_rx_set_profiling(self._h, int(bool(enable))))

    def kernel_ms(self):
        """(sync_ms, demod_ms) of the last demod_frames call (needs set_profiling(True))."""
        a, b = C.c_float(), C.c_float()
        check(self.lib.ofdm_rx_get_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def set_sync_search(self, exhaustive: bool) -> bool:
        """True if the screened sync search is active afterwards (False: the exhaustive, trial-by-trial one)."""
        return bool(check(self.lib.ofdm_rx_set_sync_search(self._h, int(bool(exhaustive)))))

    def set_max_trials(self, n: int):
        check(self.lib.ofdm_rx_set_max_trials(self._h, int(n)))

    def frame_state(self, frame: int):
        c = self.cfg
        H = np.zeros(c.nfft, np.complex64)
        g = np.zeros(c.num_data_bins, np.complex64)
        ht = np.zeros(c.nfft, np.complex64)
        check(self.lib.ofdm_rx_get_frame_state(self._h, int(frame), ptr(H), ptr(g), ptr(ht)))
        return dict(chan_freq=H, gain=g, chan_time=ht)

    def demap(self, d_sym, n, modulation="QPSK", d_hard=None, d_soft0=None, d_soft1=None, stream=None):
        check(self.lib.ofdm_demap(self._h, ptr(d_sym), int(n), _mod_bits(modulation), ptr(d_hard), ptr(d_soft0),
                                  ptr(d_soft1), ptr(stream)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ofdm_rx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FoEngine:
    """CFO-search receiver handle (reference: LEGACY/gr-ofdm-rx/python/SynchEstAndFO.py): trial x candidate sync table,
    up to 100 syncs per call, one equalised data symbol per sync."""

    def __init__(self, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr, rotators, device=0,
                 spread_code=None):
        """rotators: [len(fo_range)][nfft] candidate rotators, or None = no carrier-offset stage (gr-RXOFDM table mode).
        spread_code: None = SynchEstAndFO; a complex sequence of length DSSS = SynchEstFOAndDSSS (despread output)."""
        self.lib = _lib.load()
        rot = None if rotators is None else np.ascontiguousarray(rotators, dtype=np.complex64)
        if rot is not None and (rot.ndim != 2 or rot.shape[1] != int(nfft) or rot.shape[0] < 1):
            raise ValueError("rotators must be [len(fo_range) >= 1][nfft]")
        code = None if spread_code is None else np.ascontiguousarray(spread_code, dtype=np.complex64).ravel()
        self.dsss = 0 if code is None else int(code.size)
        self.n_spread = int(num_data_bins) // self.dsss if self.dsss else 0
        self.cfg = _lib.FoCfg(int(num_ofdm_symb), int(nfft), int(cp_len), int(num_synch_bins), int(synch_dat[0]),
                              int(synch_dat[1]), int(num_data_bins), 1 if rot is None else int(rot.shape[0]), float(snr),
                              None if rot is None else rot.ctypes.data, int(device), self.dsss,
                              None if code is None else code.ctypes.data)
        h = C.c_void_p()
        check(self.lib.ofdm_fo_create(C.byref(self.cfg), C.byref(h)))
        self.cfg.rotators = None          # the library copied the tables
        self.cfg.spread_code = None
        self._h = h
        self.report = _lib.FoReport()

    def work(self, in0: np.ndarray, out: np.ndarray) -> int:
        """Raw status (see RxEngine.work)."""
        in0 = np.ascontiguousarray(in0, dtype=np.complex64)
        if out.dtype != np.complex64 or not out.flags.c_contiguous:
            tmp = np.ascontiguousarray(out, dtype=np.complex64)
            rc = self.lib.ofdm_fo_work(self._h, ptr(in0), in0.size, ptr(tmp), tmp.size, C.byref(self.report))
            if rc >= 0:
                out[...] = tmp
            return int(rc)
        return int(self.lib.ofdm_fo_work(self._h, ptr(in0), in0.size, ptr(out), out.size, C.byref(self.report)))

    def state(self):
        c = self.cfg
        R = _lib.FO_MAX_SYNC
        mm = c.synch_S * c.num_synch_bins
        tsr = np.zeros((R, 3), np.float64)
        H = np.zeros((R, c.nfft), np.complex64)
        ht = np.zeros((R, c.nfft), np.complex64)
        esf = np.zeros((R, mm), np.complex64)
        edf = np.zeros((R, c.num_data_bins), np.complex64)
        eqg = np.zeros(c.num_synch_bins, np.complex64)
        check(self.lib.ofdm_fo_get_state(self._h, ptr(tsr), ptr(H), ptr(ht), ptr(esf), ptr(edf), ptr(eqg)))
        return dict(time_synch_ref=tsr, chan_freq=H, chan_time=ht, synch_freq=esf, data_freq=edf, eq_gain=eqg)

    def despread(self):
        out = np.zeros((_lib.FO_MAX_SYNC, self.n_spread), np.complex64)
        check(self.lib.ofdm_fo_get_despread(self._h, ptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ofdm_fo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TrkEngine:
    """Device primitives of the regression-tracking receiver (reference: LEGACY/gr-ofdm-rx/python/SynchronizeAndEstimate.py):
    strided / single sync trials, LS estimate per accepted sync, data stage.  The sequential pointer logic lives in the
    block mirror (`blocks.SynchronizeAndEstimate`)."""

    def __init__(self, nfft, cp_len, num_synch_bins, num_data_bins, synch_D, rows_sync, rows_data, snr, zc_root=23, device=0):
        self.lib = _lib.load()
        self.cfg = _lib.TrkCfg(int(nfft), int(cp_len), int(num_synch_bins), int(num_data_bins), int(synch_D), int(rows_sync),
                               int(rows_data), int(zc_root), float(snr), int(device), 0)
        h = C.c_void_p()
        check(self.lib.ofdm_trk_create(C.byref(self.cfg), C.byref(h)))
        self._h = h

    def load(self, in0: np.ndarray):
        in0 = np.ascontiguousarray(in0, dtype=np.complex64)
        check(self.lib.ofdm_trk_load(self._h, ptr(in0), in0.size))

    def trials(self, first_ptr: int, step: int, count: int):
        peak = np.zeros(max(count, 1), np.float32)
        lag = np.zeros(max(count, 1), np.int32)
        check(self.lib.ofdm_trk_trials(self._h, int(first_ptr), int(step), int(count), ptr(peak), ptr(lag)))
        return peak[:count], lag[:count]

    def accept(self, row: int, window_ptr: int, lag_sync: int, lag_data: int):
        check(self.lib.ofdm_trk_accept(self._h, int(row), int(window_ptr), int(lag_sync), int(lag_data)))

    def demod(self, ptrs, guards):
        n = len(ptrs)
        p = np.ascontiguousarray(ptrs, dtype=np.int64)
        g = np.ascontiguousarray(guards, dtype=np.uint8)
        last = np.zeros(self.cfg.num_data_bins, np.complex64)
        row = C.c_int32(-1)
        check(self.lib.ofdm_trk_demod(self._h, n, ptr(p) if n else None, ptr(g) if n else None, ptr(last), C.byref(row)))
        return int(row.value), last

    def state(self):
        c = self.cfg
        H = np.zeros((c.rows_sync, c.nfft), np.complex64)
        imp = np.zeros((c.rows_sync, c.nfft), np.complex64)
        esf = np.zeros((c.rows_sync, c.num_synch_bins), np.complex64)
        edf = np.zeros((max(c.rows_data, 0), c.num_data_bins), np.complex64)
        check(self.lib.ofdm_trk_get_state(self._h, ptr(H), ptr(imp), ptr(esf), ptr(edf) if c.rows_data > 0 else None))
        return dict(chan_freq=H, chan_impulse=imp, synch_freq=esf, data_freq=edf)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ofdm_trk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TxEngine:
    """Transmit chain handle (bit map, resource grid + ZC sync symbols, IFFT + CP + normalise) and channel."""

    def __init__(self, nfft, cp_len, num_synch_bins, num_data_bins, synch_dat=(1, 3), modulation="QPSK",
                 zc_root=23, device=0):
        self.lib = _lib.load()
        self.cfg = _lib.TxCfg(int(nfft), int(cp_len), int(num_synch_bins), int(num_data_bins), int(synch_dat[0]),
                              int(synch_dat[1]), _mod_bits(modulation), int(zc_root), int(device), 0)
        h = C.c_void_p()
        check(self.lib.ofdm_tx_create(C.byref(self.cfg), C.byref(h)))
        self._h = h

    def data_symbols(self, n_sym: int) -> int:
        c = self.cfg
        sd = c.synch_S + c.synch_D
        n = (n_sym // sd) * c.synch_D
        rem = n_sym % sd
        return n + max(0, rem - c.synch_S)

    def bits_per_frame(self, n_sym: int) -> int:
        return self.data_symbols(n_sym) * self.cfg.num_data_bins * self.cfg.modulation

    def modulate_frames(self, d_bits, n_frames, n_sym, d_iq, frame_stride=None, bits_mode=BITS_UNPACKED, stream=None):
        L = self.cfg.nfft + self.cfg.cp_len
        if frame_stride is None:
            frame_stride = n_sym * L
        check(self.lib.ofdm_tx_modulate_frames(self._h, ptr(d_bits), int(bits_mode), int(n_frames), int(n_sym),
                                               ptr(d_iq), int(frame_stride), ptr(stream)))

    # ---- decomposed stages (device buffers)
    def random_bits(self, seed, offset, d_bits, n_bits, stream=None):
        check(self.lib.ofdm_tx_random_bits(self._h, int(seed), int(offset), ptr(d_bits), int(n_bits), ptr(stream)))

    def map(self, d_bits, n_symbols, d_sym, bits_mode=BITS_UNPACKED, stream=None):
        check(self.lib.ofdm_tx_map(self._h, ptr(d_bits), int(bits_mode), int(n_symbols), ptr(d_sym), ptr(stream)))

    def set_pilots(self, locations, value=1.0 + 0.0j):
        loc = np.ascontiguousarray(list(locations), dtype=np.int32)
        check(self.lib.ofdm_tx_set_pilots(self._h, ptr(loc) if loc.size else None, int(loc.size), float(np.real(value)), float(np.imag(value))))

    def grid(self, d_sym, n_rows, d_grid, stream=None):
        check(self.lib.ofdm_tx_grid(self._h, ptr(d_sym), int(n_rows), ptr(d_grid), ptr(stream)))

    def ifft_cp(self, d_in, n_rows, d_out, do_ifft=True, add_cp=True, stream=None):
        check(self.lib.ofdm_tx_ifft_cp(self._h, ptr(d_in), int(n_rows), int(bool(do_ifft)), int(bool(add_cp)), ptr(d_out), ptr(stream)))

    def mux_symbols(self, n_data_sym: int) -> int:
        c = self.cfg
        full, rem = divmod(int(n_data_sym), c.synch_D)
        return full * (c.synch_S + c.synch_D) + (c.synch_S + rem if rem else 0)

    def mux(self, d_data, n_data_sym, d_out, stream=None) -> int:
        return int(check(self.lib.ofdm_tx_mux(self._h, ptr(d_data), int(n_data_sym), ptr(d_out), ptr(stream))))

    def sync_symbol(self) -> np.ndarray:
        c = self.cfg
        out = np.zeros((c.synch_S, c.nfft + c.cp_len), np.complex64)
        check(self.lib.ofdm_tx_get_sync_symbol(self._h, ptr(out)))
        return out

    def channel(self, d_in, n_frames, in_stride, in_len, d_taps, n_taps, d_out, out_stride, out_len,
                noise_var=0.0, seed=0, per_frame_taps=False, stream=None):
        check(self.lib.ofdm_channel_apply(self._h, ptr(d_in), int(n_frames), int(in_stride), int(in_len), ptr(d_taps),
                                          int(n_taps), int(bool(per_frame_taps)), float(noise_var), int(seed),
                                          ptr(d_out), int(out_stride), int(out_len), ptr(stream)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ofdm_tx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
