"""GNU Radio module `TXOFDM` (reference: gr-TXOFDM/python/__init__.py), MI355X-native.
The reference's __init__ imports a class name that does not exist in its own file (ImportError); here it works."""
from ofdm_mi355x.blocks import tx_signal_transmitter  # noqa: F401
