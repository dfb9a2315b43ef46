"""GNU Radio module `OFDMReceiver` (reference: LEGACY/gr-ofdm-rx, grc/OFDMReceiver_BitRecovery.block.yml), MI355X-native."""
from ofdm_mi355x.blocks import BitRecovery, SynchEstAndFO, SynchEstFOAndDSSS, SynchronizeAndEstimate  # noqa: F401
from ofdm_mi355x.blocks import LegacySynchAndChanEst as SynchAndChanEst  # noqa: F401
