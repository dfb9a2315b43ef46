"""GNU Radio module `utsa_ofdm` (reference: gr-utsa_ofdm/python/__init__.py:23-24), MI355X-native."""
from ofdm_mi355x.blocks import SynchAndChanEst, TxSignalTransmitter  # noqa: F401
