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
         make_args=["num_ofdm_symb", "nfft", "cp_len", "num_synch_bins", "synch_dat", "num_data_bins", "snr",
                    "scale_factor_gate", "directory_name", "file_name_cest", "diagnostics", "genie", "channel"],
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for b in BLOCKS:
        with open(os.path.join(args.out, b["id"] + ".block.yml"), "w") as f:
            f.write(block_yaml(b))
        print("wrote", b["id"] + ".block.yml")


if __name__ == "__main__":
    main()
