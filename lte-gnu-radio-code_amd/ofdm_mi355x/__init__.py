"""MI355X-native OFDM PHY hot path: Python host side of libofdm_mi355x.so."""
from ._lib import (BITS_NONE, BITS_PACKED, BITS_UNPACKED, COMPAT_RXOFDM, COMPAT_UTSA, LIB_PATH, OfdmError,  # noqa: F401
                   OfdmLibraryError, load)
from .engine import DeviceBuffer, FoEngine, RxEngine, TrkEngine, TxEngine, bins_p, count_bit_errors, zadoff_chu  # noqa: F401
from .safe_pickle import UnsafePickleError, load_ndarray  # noqa: F401
