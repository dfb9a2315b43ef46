"""ctypes binding of libofdm_mi355x.so (C ABI: include/ofdm_mi355x.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded, importing
any compute entry point raises ``OfdmLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFDM_MI355X_LIB", os.path.join(_HERE, "libofdm_mi355x.so"))

OFDM_OK = 0
OFDM_ERR_INVALID = -1
OFDM_ERR_HIP = -2
OFDM_ERR_INDEX = -3
OFDM_ERR_SHAPE = -4
OFDM_ERR_NOMEM = -5
OFDM_ERR_UNBOUND = -6

COMPAT_UTSA = 0
COMPAT_RXOFDM = 1
BITS_NONE, BITS_PACKED, BITS_UNPACKED = 0, 1, 2
MODULATION_BITS = {"BPSK": 1, "QPSK": 2, "16QAM": 4, "64QAM": 6}


class OfdmLibraryError(RuntimeError):
    pass


class OfdmError(RuntimeError):
    pass


class RxCfg(C.Structure):
    _fields_ = [("num_ofdm_symb", C.c_int32), ("nfft", C.c_int32), ("cp_len", C.c_int32),
                ("num_synch_bins", C.c_int32), ("synch_S", C.c_int32), ("synch_D", C.c_int32),
                ("num_data_bins", C.c_int32), ("snr", C.c_double), ("scale_factor_gate", C.c_double),
                ("compat", C.c_int32), ("modulation", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32)]


class RxReport(C.Structure):
    _fields_ = [("time_synch_ref", C.c_double * 3), ("detected", C.c_int32), ("trials_run", C.c_int32),
                ("count", C.c_int32), ("corr_obs", C.c_int32), ("n_data_items", C.c_int64)]


class TxCfg(C.Structure):
    _fields_ = [("nfft", C.c_int32), ("cp_len", C.c_int32), ("num_synch_bins", C.c_int32),
                ("num_data_bins", C.c_int32), ("synch_S", C.c_int32), ("synch_D", C.c_int32),
                ("modulation", C.c_int32), ("zc_root", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32)]


class FoCfg(C.Structure):
    _fields_ = [("num_ofdm_symb", C.c_int32), ("nfft", C.c_int32), ("cp_len", C.c_int32),
                ("num_synch_bins", C.c_int32), ("synch_S", C.c_int32), ("synch_D", C.c_int32),
                ("num_data_bins", C.c_int32), ("n_fo", C.c_int32), ("snr", C.c_double),
                ("rotators", C.c_void_p), ("device", C.c_int32), ("dsss", C.c_int32), ("spread_code", C.c_void_p)]


class FoReport(C.Structure):
    _fields_ = [("n_sync", C.c_int32), ("count", C.c_int32), ("dmax_tmp_ind", C.c_int32), ("trials_run", C.c_int32),
                ("n_data_items", C.c_int64)]


class TrkCfg(C.Structure):
    _fields_ = [("nfft", C.c_int32), ("cp_len", C.c_int32), ("num_synch_bins", C.c_int32), ("num_data_bins", C.c_int32),
                ("synch_D", C.c_int32), ("rows_sync", C.c_int32), ("rows_data", C.c_int32), ("zc_root", C.c_int32),
                ("snr", C.c_double), ("device", C.c_int32), ("reserved", C.c_int32)]


FO_MAX_SYNC = 100

# name -> (restype, argtypes): exactly the prototypes of include/ofdm_mi355x.h
PROTOTYPES = {
    "ofdm_abi_version": (C.c_int, []),
    "ofdm_shard_frames": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ofdm_last_error": (C.c_char_p, []),
    "ofdm_device_malloc": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.c_int64]),
    "ofdm_device_free": (C.c_int, [C.c_int32, C.c_void_p]),
    "ofdm_memcpy_h2d": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    "ofdm_memcpy_d2h": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    "ofdm_device_synchronize": (C.c_int, [C.c_int32]),
    "ofdm_bandwidth_probe": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int64, C.c_void_p]),
    "ofdm_count_bit_errors": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ofdm_rx_create": (C.c_int, [C.POINTER(RxCfg), C.POINTER(C.c_void_p)]),
    "ofdm_rx_destroy": (C.c_int, [C.c_void_p]),
    "ofdm_rx_work": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(RxReport)]),
    "ofdm_rx_get_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ofdm_rx_demod_frames": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ofdm_rx_get_frame_state": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ofdm_rx_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "ofdm_rx_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "ofdm_rx_get_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "ofdm_rx_set_max_trials": (C.c_int, [C.c_void_p, C.c_int32]),
    "ofdm_rx_set_sync_search": (C.c_int, [C.c_void_p, C.c_int32]),
    "ofdm_demap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ofdm_fo_create": (C.c_int, [C.POINTER(FoCfg), C.POINTER(C.c_void_p)]),
    "ofdm_fo_destroy": (C.c_int, [C.c_void_p]),
    "ofdm_fo_work": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(FoReport)]),
    "ofdm_fo_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ofdm_fo_get_despread": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ofdm_trk_create": (C.c_int, [C.POINTER(TrkCfg), C.POINTER(C.c_void_p)]),
    "ofdm_trk_destroy": (C.c_int, [C.c_void_p]),
    "ofdm_trk_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "ofdm_trk_trials": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ofdm_trk_accept": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32]),
    "ofdm_trk_demod": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "ofdm_trk_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ofdm_tx_create": (C.c_int, [C.POINTER(TxCfg), C.POINTER(C.c_void_p)]),
    "ofdm_tx_destroy": (C.c_int, [C.c_void_p]),
    "ofdm_tx_modulate_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
                                          C.c_int64, C.c_void_p]),
    "ofdm_tx_random_bits": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p]),
    "ofdm_tx_map": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "ofdm_tx_set_pilots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float]),
    "ofdm_tx_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ofdm_tx_ifft_cp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ofdm_tx_mux": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ofdm_tx_get_sync_symbol": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ofdm_channel_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int32,
                                     C.c_int32, C.c_float, C.c_uint64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
}

_lib = None
hip_runtime_path = None     # the libamdhip64 this process ended up with (diagnostic)


def _trace(what):
    """OFDM_MI355X_TRACE_LOAD=1: one time-stamped line on stderr per step of load() (which dlopen a stuck start-up sits in)."""
    if os.environ.get("OFDM_MI355X_TRACE_LOAD") == "1":
        import sys
        import time
        print("[ofdm_mi355x.load %.3f] %s" % (time.time(), what), file=sys.stderr, flush=True)


def _mapped_hip_runtimes():
    """Paths of every libamdhip64 mapped into this process (Linux)."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if "libamdhip64" in path and path not in found:
                    found.append(path)
    except OSError:
        pass
    return found


def _pin_hip_runtime():
    """One HIP runtime per process, whatever the import order -- and torch's device code registered BEFORE that runtime starts.

    libofdm_mi355x.so needs `libamdhip64.so.7`; the PyTorch-ROCm wheel ships its own library under the SAME soname.  The
    dynamic linker keeps whichever was mapped first for both users.  The system runtime first and torch second leaves torch
    on a runtime it was not built for (`RuntimeError: No HIP GPUs are available` on the first torch.cuda call); torch's
    runtime first works for both.

    Mapping torch's runtime first is not enough, though (round 3, `profiles/r03_import_order_library_first_STUCK.log`): when this
    library has INITIALISED the runtime and `import torch` comes afterwards, `from torch._C import *` (torch/__init__.py) loads
    ~1 GB of shared objects whose constructors register their device code with a runtime that is already live -- it is unpacked on
    the spot instead of lazily.  That takes 6-8 s on a good day and was caught sitting there for 150 s (the watchdog's stack dump).
    So when torch is installed and not imported yet, it is imported HERE, before anything touches the GPU: its device code is
    registered lazily, as in every torch-first program (about 1.5 s once per process).  OFDM_MI355X_NO_TORCH_IMPORT=1 keeps torch
    out (then only its runtime is mapped first, as before); OFDM_MI355X_SYSTEM_HIP=1 opts out of both.  Hosts without torch
    (GNU Radio, the C example) get the system runtime."""
    global hip_runtime_path
    mapped = _mapped_hip_runtimes()
    if mapped:
        hip_runtime_path = mapped[0]
        return
    if os.environ.get("OFDM_MI355X_SYSTEM_HIP") == "1":
        return
    import importlib.util
    import sys
    try:
        spec = sys.modules["torch"].__spec__ if "torch" in sys.modules else importlib.util.find_spec("torch")
    except (ImportError, ValueError, AttributeError):
        spec = None
    if spec is None or not spec.origin:
        return
    if "torch" not in sys.modules and os.environ.get("OFDM_MI355X_NO_TORCH_IMPORT") != "1":
        _trace("importing torch before the HIP runtime starts")
        try:
            import importlib
            importlib.import_module("torch")
        except Exception as e:                              # a broken torch install must not take this library down with it
            _trace("import torch failed (%s: %s): mapping its runtime only" % (type(e).__name__, e))
        mapped = _mapped_hip_runtimes()
        if mapped:
            hip_runtime_path = mapped[0]
            return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            hip_runtime_path = cand
        except OSError:
            pass


def load():
    """Load the shared library (once).  Raises OfdmLibraryError when it is absent: there is no fallback."""
    global _lib, hip_runtime_path
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OfdmLibraryError(
            "HIP library %s not found: build it with `make -C lte-gnu-radio-code_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback." % LIB_PATH)
    _trace("pinning the HIP runtime")
    _pin_hip_runtime()
    _trace("HIP runtime: %s; dlopen %s" % (hip_runtime_path or "(system, via DT_NEEDED)", LIB_PATH))
    try:
        lib = C.CDLL(LIB_PATH)          # CDLL calls release the GIL (GNU Radio: one thread per block)
    except OSError as e:
        raise OfdmLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    _trace("library mapped")
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise OfdmLibraryError("%s does not export %s" % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.ofdm_abi_version() != 1:
        raise OfdmLibraryError("ABI version mismatch")
    mapped = _mapped_hip_runtimes()
    if len(mapped) > 1:
        raise OfdmLibraryError("two HIP runtimes are mapped into this process (%s): import ofdm_mi355x before anything else "
                               "that loads libamdhip64, or import torch first" % ", ".join(mapped))
    if mapped:
        hip_runtime_path = mapped[0]
    _lib = lib
    return lib


def last_error() -> str:
    return load().ofdm_last_error().decode("utf-8", "replace")


def check(rc: int):
    """Map a negative ofdm_status to the exception the reference's NumPy code would raise."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == OFDM_ERR_INDEX:
        raise IndexError(msg)
    if rc == OFDM_ERR_SHAPE or rc == OFDM_ERR_INVALID:
        raise ValueError(msg)
    if rc == OFDM_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == OFDM_ERR_UNBOUND:
        raise UnboundLocalError(msg)
    raise OfdmError(msg)


def ptr(x):
    """Device/host address of a torch tensor, numpy array, DeviceBuffer, int or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if hasattr(x, "ctypes"):
        return C.c_void_p(x.ctypes.data)
    raise TypeError("cannot take the address of %r" % type(x))
