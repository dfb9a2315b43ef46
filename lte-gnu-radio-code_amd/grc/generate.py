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
TX_PARAMS = [("case", "Case", "int"), ("pickle_directory", "IQ file directory", "string"), ("pickle_file", "IQ file (.pckl/.npy)", "string")]

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
         params=[("modulation", "Modulation", "string"), ("directory_name", "Directory", "string"), ("diagnostics", "Diagnostics", "int")],
         make_args=["modulation", "directory_name", "diagnostics"], inputs=CPLX_IN, outputs=None),
    dict(id="OFDMReceiver_SynchAndChanEst", label="SynchAndChanEst (MI355X)", category="[OFDMReceiver]", module="OFDMReceiver",
         cls="SynchAndChanEst",
         params=[("num_ofdm_symb", "No. of OFDM Symbols", "int"), ("nfft", "FFT Size", "int"), ("cp_len", "CP Length", "int"),
                 ("num_synch_bins", "No. of Synch Bins", "int"), ("synch_dat", "Synch-Data Pattern", "raw"),
                 ("num_data_bins", "No. of Data Bins", "int"), ("SNR", "SNR", "float"), ("directory_name", "Directory Path", "string"),
                 ("file_name_cest", "Var: Chan Est -- File Name", "string"), ("diagnostics", "Diagnostics", "int")],
         make_args=["num_ofdm_symb", "nfft", "cp_len", "num_synch_bins", "synch_dat", "num_data_bins", "SNR", "directory_name",
                    "file_name_cest", "diagnostics"], inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="OFDMReceiver_SynchronizeAndEstimate", label="SynchronizeAndEstimate (MI355X)", category="[OFDMReceiver]",
         module="OFDMReceiver", cls="SynchronizeAndEstimate", params=[("case", "Case", "int")], make_args=["case"],
         inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="OFDMReceiver_SynchEstAndFO", label="SynchEstAndFO (MI355X)", category="[OFDMReceiver]", module="OFDMReceiver",
         cls="SynchEstAndFO",
         params=[("case", "Case Number", "int"), ("fo_range", "F Offset Range", "raw"), ("directory_name", "Directory Path", "string"),
                 ("file_name_cest", "Var: Chan Est -- File Name", "string"), ("diagnostics", "Diagnostics", "int")],
         make_args=["case", "fo_range", "directory_name", "file_name_cest", "diagnostics"], inputs=CPLX_IN, outputs=CPLX_IN),
    dict(id="OFDMReceiver_SynchEstFOAndDSSS", label="SynchEstFOAndDSSS (MI355X)", category="[OFDMReceiver]", module="OFDMReceiver",
         cls="SynchEstFOAndDSSS",
         params=[("case", "Case Number", "int"), ("fo_range", "F Offset Range", "raw"), ("directory_name", "Directory Path", "string"),
                 ("file_name_cest", "Var: Chan Est -- File Name", "string"), ("diagnostics", "Diagnostics", "int")],
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


def block_yaml(b):
    doc = dict(id=b["id"], label=b["label"], category=b["category"],
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
