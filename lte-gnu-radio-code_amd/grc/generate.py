#!/usr/bin/env python3
"""Emit the GRC block descriptions (*.block.yml) of the drop-in modules.

    python generate.py --out ~/.local/share/gnuradio/grc/blocks     (any directory on GRC's block path)

The block ids, parameter ids and `make:` templates are the drop-in contract with existing .grc flowgraphs
(e.g. `ofdm_chain.grc` references RXOFDM_synch_and_chan_est / TXOFDM_tx_signal_transmitter), so they are
kept identical to the reference's; everything is produced from the table below, nothing is copied.
"""
import argparse
import os

import yaml

CPLX_IN = [dict(domain="stream", dtype="complex")]
BYTE_IN = [dict(domain="stream", dtype="byte")]

RX_PARAMS = [("num_ofdm_symb", "OFDM symbols per buffer", "int"), ("nfft", "FFT size", "int"),
             ("cp_len", "Cyclic prefix length", "int"), ("num_synch_bins", "Sync bins", "int"),
             ("synch_dat", "[sync, data] symbol pattern", "raw"), ("num_data_bins", "Data bins", "int"),
             ("snr", "SNR", "int")]
FILE_PARAMS = [("directory_name", "Output directory / prefix", "string"), ("file_name_cest", "Channel-estimate file", "string")]
TX_PARAMS = [("case", "Numerology case", "int"), ("pickle_directory", "Directory of the recorded IQ frame", "string"), ("pickle_file", "Recorded IQ frame (.pckl / .npy)", "string")]

BLOCKS = [
    dict(id="utsa_ofdm_SynchAndChanEst", label="SynchAndChanEst (MI355X)", category="[utsa_ofdm]", module="utsa_ofdm",
         cls="SynchAndChanEst",
         params=RX_PARAMS + [("scale_factor_gate", "Correlation gate", "float")] + FILE_PARAMS +
         [("diagnostics", "Diagnostics", "bool"), ("genie", "Genie", "bool"), ("channel", "Channel type (genie only)", "string")],
         # like the reference's template, `channel` is declared but not passed (the constructor defaults it)
         make_args=["num_ofdm_symb", "nfft", "cp_len", "num_synch_bins", "synch_dat", "num_data_bins", "snr",
                    "scale_factor_gate", "directory_name", "file_name_cest", "diagnostics", "genie"],
         inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="utsa_ofdm_TxSignalTransmitter", label="TxSignalTransmitter (MI355X)", category="[utsa_ofdm]",
         module="utsa_ofdm", cls="TxSignalTransmitter", params=TX_PARAMS,
         make_args=["pickle_directory", "pickle_file"], inputs=None, outputs=CPLX_IN),
    dict(id="RXOFDM_synch_and_chan_est", label="synch_and_chan_est (MI355X)", category="[RXOFDM]", module="RXOFDM",
         cls="synch_and_chan_est",
         params=RX_PARAMS + FILE_PARAMS + [("diagnostics", "Diagnostics", "int"), ("genie", "Genie", "int")],
         make_args=["num_ofdm_symb", "nfft", "cp_len", "num_synch_bins", "synch_dat", "num_data_bins", "snr",
                    "directory_name", "file_name_cest", "diagnostics", "genie"],
         inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="TXOFDM_tx_signal_transmitter", label="tx_signal_transmitter (MI355X)", category="[OFDM Transmitter]",
         module="TXOFDM", cls="tx_signal_transmitter", params=TX_PARAMS,
         make_args=["case", "pickle_directory", "pickle_file"], inputs=None, outputs=CPLX_IN),
    dict(id="OFDMReceiver_BitRecovery", label="Bit Recovery (MI355X)", category="[OFDMReceiver]", module="OFDMReceiver",
         cls="BitRecovery",
         params=[("modulation", "Constellation (BPSK / QPSK / 16QAM / 64QAM)", "string"), ("directory_name", "Output directory / prefix", "string"), ("diagnostics", "Write diagnostics (0/1)", "int")],
         make_args=["modulation", "directory_name", "diagnostics"], inputs=CPLX_IN, outputs=None),
    dict(id="OFDMReceiver_SynchAndChanEst", label="SynchAndChanEst (MI355X)", category="[OFDMReceiver]", module="OFDMReceiver",
         cls="SynchAndChanEst",
         params=[("num_ofdm_symb", "OFDM symbols per buffer", "int"), ("nfft", "FFT size", "int"), ("cp_len", "Cyclic prefix length", "int"),
                 ("num_synch_bins", "Sync bins", "int"), ("synch_dat", "[sync, data] symbol pattern", "raw"),
                 ("num_data_bins", "Data bins", "int"), ("SNR", "SNR", "float"), ("directory_name", "Output directory / prefix", "string"),
                 ("file_name_cest", "Channel-estimate file", "string"), ("diagnostics", "Write diagnostics (0/1)", "int")],
         make_args=["num_ofdm_symb", "nfft", "cp_len", "num_synch_bins", "synch_dat", "num_data_bins", "SNR", "directory_name",
                    "file_name_cest", "diagnostics"], inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="OFDMReceiver_SynchronizeAndEstimate", label="SynchronizeAndEstimate (MI355X)", category="[OFDMReceiver]",
         module="OFDMReceiver", cls="SynchronizeAndEstimate", params=[("case", "Numerology case", "int")], make_args=["case"],
         inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="OFDMReceiver_SynchEstAndFO", label="SynchEstAndFO (MI355X)", category="[OFDMReceiver]", module="OFDMReceiver",
         cls="SynchEstAndFO",
         params=[("case", "Numerology case", "int"), ("fo_range", "Carrier-offset candidates [Hz]", "raw"), ("directory_name", "Output directory / prefix", "string"),
                 ("file_name_cest", "Channel-estimate file", "string"), ("diagnostics", "Write diagnostics (0/1)", "int")],
         make_args=["case", "fo_range", "directory_name", "file_name_cest", "diagnostics"], inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="OFDMReceiver_SynchEstFOAndDSSS", label="SynchEstFOAndDSSS (MI355X)", category="[OFDMReceiver]", module="OFDMReceiver",
         cls="SynchEstFOAndDSSS",
         params=[("case", "Numerology case", "int"), ("fo_range", "Carrier-offset candidates [Hz]", "raw"), ("directory_name", "Output directory / prefix", "string"),
                 ("file_name_cest", "Channel-estimate file", "string"), ("diagnostics", "Write diagnostics (0/1)", "int")],
         make_args=["case", "fo_range", "directory_name", "file_name_cest", "diagnostics"], inputs=CPLX_IN, outputs=CPLX_IN),
    # ---- the decomposed live transmitter: block ids and parameter ids as the reference's flowgraph uses them
    # (LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc:701-975); the reference ships no block description for them
    dict(id="txOFDM_random_bit_source", label="Random Bit Source (MI355X)", category="[txOFDM]", module="txOFDM",
         cls="random_bit_source", params=[], make_args=[], inputs=None, outputs=BYTE_IN),
    dict(id="txOFDM_ConstellationModulation", label="Constellation Modulation (MI355X)", category="[txOFDM]", module="txOFDM",
         cls="ConstellationModulation", params=[("modulation", "Modulation", "string")], make_args=["modulation"],
         inputs=BYTE_IN, outputs=CPLX_IN),
    dict(id="txOFDM_OFDM_Modulation", label="OFDM Modulation (MI355X)", category="[txOFDM]", module="txOFDM",
         cls="OFDM_Modulation", params=[("fft_size", "FFT Size", "int"), ("pilot_locations", "Pilot Locations", "raw")],
         make_args=["fft_size", "pilot_locations"], inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="txOFDM_IFFT", label="IFFT (MI355X)", category="[txOFDM]", module="txOFDM", cls="IFFT",
         params=[("fft_size", "FFT Size", "int")], make_args=["fft_size"], inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="txOFDM_CyclicPrefix", label="Cyclic Prefix (MI355X)", category="[txOFDM]", module="txOFDM", cls="CyclicPrefix",
         params=[("fft_size", "FFT Size", "int"), ("cp_size", "CP Size", "int")], make_args=["fft_size", "cp_size"],
         inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="txOFDM_SynchDataMux", label="Synch / Data Mux (MI355X)", category="[txOFDM]", module="txOFDM", cls="SynchDataMux",
         params=[("fft_size", "FFT Size", "int"), ("cp_size", "CP Size", "int"), ("prime_no", "Zadoff-Chu root", "int"),
                 ("synch_every", "Data symbols per sync symbol", "int"), ("synch_length", "Sync bins", "int")],
         make_args=["fft_size", "cp_size", "prime_no", "synch_every", "synch_length"], inputs=CPLX_IN, outputs=CPLX_IN),
]

# What each block does and what runs behind it (emitted as the block's `documentation:`; GRC shows it in the block's
# properties dialog).  Reference behaviour is cited as file:line of tayloreisman16/LTE-GNU-Radio-Code.
DOCS = {
    "utsa_ofdm_SynchAndChanEst": "Receiver of ofdm_chain.py: Zadoff-Chu timing search, least-squares channel estimate on the sync symbol, then CP "
        "strip + FFT + data-bin gather + one-tap MMSE equaliser for every data symbol of the buffer (gr-utsa_ofdm/python/SynchAndChanEst.py:"
        "139-262).  work() copies the buffer to the GPU and runs ofdm_rx_work of libofdm_mi355x.so: one screened sync-search launch "
        "and one demod launch (hand-written HIP, gfx950).  Output items: equalised data-bin symbols, emitted from the second call on.",
    "utsa_ofdm_TxSignalTransmitter": "Replays a recorded OFDM frame (the reference's tx_data_* pickle, parsed without unpickling, or a .npy) as a "
        "cyclic complex stream (gr-utsa_ofdm/python/TxSignalTransmitter.py).  Host memcpy only; the modulator that produces such frames is "
        "ofdm_mi355x.TxEngine / the txOFDM blocks.",
    "RXOFDM_synch_and_chan_est": "gr-RXOFDM flavour of the receiver block (stride cp-1 sync search, ZC root and bin conventions of "
        "gr-RXOFDM/python/synch_and_chan_est.py); table_mode=True keeps one estimate per detected sync symbol.  Same HIP engine as "
        "utsa_ofdm_SynchAndChanEst in its RXOFDM compatibility mode.",
    "TXOFDM_tx_signal_transmitter": "gr-TXOFDM flavour of the frame replay source (case selects the numerology).  Host memcpy only.",
    "OFDMReceiver_BitRecovery": "Hard / soft bit decisions on equalised symbols (LEGACY/gr-ofdm-rx/python/BitRecovery.py:45-148): nearest "
        "constellation point with the reference's tie rule, max-log soft metrics with the noise estimate of the buffer.  HIP de-map "
        "kernels behind ofdm_demap; 16/64-QAM are an extension (TS 36.211 Gray maps).",
    "OFDMReceiver_SynchAndChanEst": "Legacy receiver that keeps a table of every sync symbol found in the buffer "
        "(LEGACY/gr-ofdm-rx/python/SynchAndChanEst.py).  Same HIP engine, legacy compatibility mode.",
    "OFDMReceiver_SynchronizeAndEstimate": "Legacy receiver whose sync-window pointer follows a straight line fitted through the last five sync "
        "positions (LEGACY/gr-ofdm-rx/python/SynchronizeAndEstimate.py:209-343).  Window evaluation, estimate and demod on the GPU "
        "(ofdm_trk_*), pointer logic on the host (ofdm_mi355x/tracker.py).",
    "OFDMReceiver_SynchEstAndFO": "Legacy receiver with a carrier-frequency-offset search: every sync window is evaluated under each candidate "
        "rotator and the strongest wins (LEGACY/gr-ofdm-rx/python/SynchEstAndFO.py:196-339).  One batched launch evaluates all trials x "
        "candidates (ofdm_fo_*).",
    "OFDMReceiver_SynchEstFOAndDSSS": "The CFO-search receiver followed by direct-sequence despreading across the data bins "
        "(LEGACY/gr-ofdm-rx/python/SynchEstFOAndDSSS.py:253-262,391-399).",
    "txOFDM_random_bit_source": "Source of uniform random bits, one per byte, from a counter-based Philox4x32-10 stream (any window can be "
        "regenerated).  Block id from LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc; the reference ships no code for the txOFDM module.",
    "txOFDM_ConstellationModulation": "Bits to constellation points, MSB first (MultiAntennaSystem.py:150-178; TS 36.211 maps for 16/64-QAM).",
    "txOFDM_OFDM_Modulation": "Rows of data symbols onto the occupied bins of an fft_size resource grid; pilot_locations are signed bin "
        "offsets that carry the pilot value (MultiAntennaSystem.py:135-139,182-183).",
    "txOFDM_IFFT": "numpy.fft.ifft per row of fft_size bins, on the register/LDS radix-16 FFT of csrc/fft_core.hpp.",
    "txOFDM_CyclicPrefix": "Prepends the last cp_size samples and applies the modulator's per-symbol power normalisation "
        "(MultiAntennaSystem.py:200-218).",
    "txOFDM_SynchDataMux": "Inserts one Zadoff-Chu sync symbol (root prime_no on synch_length bins) before every synch_every data symbols.",
}


def block_yaml(b):
    doc = dict(id=b["id"], label=b["label"], category=b["category"], documentation=DOCS[b["id"]],
               parameters=[dict(id=i, label=l, dtype=t) for i, l, t in b["params"]],
               templates=dict(imports="import " + b["module"],
                              make="%s.%s(%s)" % (b["module"], b["cls"], ", ".join("${%s}" % a for a in b["make_args"]))),
               file_format=1)
    if b["inputs"]:
        doc["inputs"] = [dict(x) for x in b["inputs"]]
    if b["outputs"]:
        doc["outputs"] = [dict(x) for x in b["outputs"]]
    return yaml.safe_dump(doc, sort_keys=False)


def write_all(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    names = []
    for b in BLOCKS:
        names.append(b["id"] + ".block.yml")
        with open(os.path.join(out_dir, names[-1]), "w") as f:
            f.write(block_yaml(b))
    return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.dirname(os.path.abspath(__file__)),
                    help="directory on GRC's block path (default: next to this script, where the generated files are committed)")
    args = ap.parse_args()
    for n in write_all(args.out):
        print("wrote", n)


if __name__ == "__main__":
    main()
