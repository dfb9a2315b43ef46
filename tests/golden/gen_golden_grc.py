#!/usr/bin/env python3
"""Records the GRC contract of the reference into tests/golden/grc_contract.json (run in the build container only; reads
/root/reference as TEXT/YAML/XML data, executes nothing from it).

 * for every reference *.block.yml that has a drop-in here: block id, parameter ids (in order), the `make:` template with
   whitespace removed, and the number of stream inputs / outputs;
 * for the `txOFDM_*` blocks, which the reference only instantiates in LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc: block id and
   the parameter keys the flowgraph sets (GRC's own bookkeeping keys removed).
"""
import json
import os
import re
import xml.etree.ElementTree as ET

import yaml

G = "/root/reference/GNU-Radio-Repositories"
YMLS = [
    "gr-utsa_ofdm/grc/utsa_ofdm_SynchAndChanEst.block.yml", "gr-utsa_ofdm/grc/utsa_ofdm_TxSignalTransmitter.block.yml",
    "gr-RXOFDM/grc/RXOFDM_synch_and_chan_est.block.yml", "gr-TXOFDM/grc/TXOFDM_tx_signal_transmitter.block.yml",
    "LEGACY/gr-ofdm-rx/grc/OFDMReceiver_BitRecovery.block.yml", "LEGACY/gr-ofdm-rx/grc/OFDMReceiver_SynchAndChanEst.block.yml",
    "LEGACY/gr-ofdm-rx/grc/OFDMReceiver_SynchronizeAndEstimate.block.yml", "LEGACY/gr-ofdm-rx/grc/OFDMReceiver_SynchEstAndFO.block.yml",
    "LEGACY/gr-ofdm-rx/grc/OFDMReceiver_SynchEstFOAndDSSS.block.yml",
]
GRC_KEYS = {"alias", "comment", "affinity", "_enabled", "_coordinate", "_rotation", "id", "maxoutbuf", "minoutbuf"}


def main():
    out = {}
    for rel in YMLS:
        d = yaml.safe_load(open(os.path.join(G, rel)))
        out[d["id"]] = dict(source=rel, params=[p["id"] for p in d.get("parameters", [])],
                            make=re.sub(r"\s+", "", d["templates"]["make"]),
                            n_inputs=len(d.get("inputs") or []), n_outputs=len(d.get("outputs") or []))
    rel = "LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc"
    root = ET.parse(os.path.join(G, rel)).getroot()
    for blk in root.findall("block"):
        key = blk.find("key").text
        if not key.startswith("txOFDM_"):
            continue
        params = sorted(p.find("key").text for p in blk.findall("param") if p.find("key").text not in GRC_KEYS)
        out[key] = dict(source=rel, params_set_by_flowgraph=params)
    conns = []
    for c in root.findall("connection"):
        a, b = c.find("source_block_id").text, c.find("sink_block_id").text
        if a.startswith("txOFDM_") and b.startswith("txOFDM_"):
            conns.append([a.rsplit("_", 1)[0], b.rsplit("_", 1)[0]])
    out["__txOFDM_connections__"] = sorted(conns)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "grc_contract.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, len(out), "entries")


if __name__ == "__main__":
    main()
