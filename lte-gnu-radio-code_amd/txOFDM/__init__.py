"""GNU Radio module `txOFDM`: the decomposed live transmitter the reference's flowgraph
LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc:701-975 instantiates (the module itself is absent from the reference), MI355X-native."""
from ofdm_mi355x.tx_blocks import (ConstellationModulation, CyclicPrefix, IFFT, OFDM_Modulation, SynchDataMux,  # noqa: F401
                                   random_bit_source)
