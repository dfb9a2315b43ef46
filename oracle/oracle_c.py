"""ctypes wrapper of oracle/libofdm_oracle_c.so (the plain-C scalar restatement, oracle/ofdm_oracle_c.c).

TEST / MEASUREMENT INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's cpu_baseline child, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libofdm_oracle_c.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.ofdm_oracle_c_rx_work.restype = C.c_int
        lib.ofdm_oracle_c_rx_work.argtypes = [C.c_void_p, C.c_int64] + [C.c_int] * 6 + [C.c_double, C.c_double, C.c_int,
                                                                                     C.c_void_p, C.c_void_p, C.c_void_p]
        assert lib.ofdm_oracle_c_version() == 1
        _lib = lib
    return _lib


def rx_work(iq, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr, gate=0.7):
    """Fresh instance, one work() call (SynchAndChanEst.py:135-262).  Returns (time_synch_ref[3], est_chan_freq_P[0],
    est_data_freq [num_ofdm_symb, Kd], trials evaluated)."""
    lib = load()
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    tsr = np.zeros(3)
    chan = np.zeros(nfft, np.complex128)
    data = np.zeros((num_ofdm_symb, num_data_bins), np.complex128)
    rc = lib.ofdm_oracle_c_rx_work(iq.ctypes.data, len(iq), nfft, cp_len, num_synch_bins, num_data_bins, int(synch_dat[0]),
                                   int(synch_dat[1]), float(snr), float(gate), num_ofdm_symb, tsr.ctypes.data, chan.ctypes.data,
                                   data.ctypes.data)
    if rc == -3:
        raise IndexError("index out of bounds for est_data_freq (num_ofdm_symb too small for the buffer)")
    if rc < 0:
        raise RuntimeError("ofdm_oracle_c_rx_work failed: %d" % rc)
    return tsr, chan, data, rc


def rx_work_frames(iq, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr, gate=0.7, n_threads=1):
    """`rx_work` over the rows of iq [n_frames, frame_len] (a fresh instance per frame) on n_threads OpenMP threads.
    Returns (time_synch_ref [n_frames, 3], frames with a detection).  Throughput leg of the CPU baseline."""
    lib = load()
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    n_frames, frame_len = iq.shape
    tsr = np.zeros((n_frames, 3))
    fn = lib.ofdm_oracle_c_rx_work_frames
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                   C.c_int, C.c_void_p, C.c_int]
    rc = fn(iq.ctypes.data, n_frames, frame_len, nfft, cp_len, num_synch_bins, num_data_bins, int(synch_dat[0]), int(synch_dat[1]),
            float(snr), float(gate), num_ofdm_symb, tsr.ctypes.data, int(n_threads))
    if rc < 0:
        raise RuntimeError("ofdm_oracle_c_rx_work_frames failed: %d" % rc)
    return tsr, rc

