"""GNU Radio module `RXOFDM` (reference: gr-RXOFDM/python/__init__.py), MI355X-native."""
from ofdm_mi355x.blocks import synch_and_chan_est, synch_and_chan_est_table  # noqa: F401
