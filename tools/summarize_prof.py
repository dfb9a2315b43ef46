#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the small summaries committed under profiles/.

  tools/summarize_prof.py TAG KT_DIR [FETCH_DIR WRITE_DIR]

Writes profiles/TAG_kernel_stats.csv (copy of rocprofv3's --stats table) and, when PMC directories are
given, profiles/TAG_pmc_traffic.json with per-kernel FETCH_SIZE / WRITE_SIZE means (KB) and the HBM bytes
per launch after the gfx950 correction prescribed by MI355X_MICROARCH.md (FETCH_SIZE counts 1/2 of a wide
coalesced stream read; WRITE_SIZE is exact) -- the correction is re-validated in the same run on
channel_kernel, whose byte count is known exactly (8 B/lane coalesced read + write of 256 x frame_len samples).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc(dirname, counter):
    f = glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), max(v), len(v)) for k, v in agg.items()}


def main():
    tag, kt = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    ks = glob.glob(os.path.join(kt, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(ks, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    if len(sys.argv) >= 5:
        fe, wr = pmc(sys.argv[3], "FETCH_SIZE"), pmc(sys.argv[4], "WRITE_SIZE")
        out = {}
        for k in sorted(set(fe) | set(wr)):
            if "ofdm::" not in k:
                continue
            f_mean, f_max, n = fe.get(k, (0, 0, 0))
            w_mean, w_max, _ = wr.get(k, (0, 0, 0))
            out[k] = dict(launches=n, FETCH_SIZE_KB_mean=round(f_mean, 1), FETCH_SIZE_KB_max=round(f_max, 1),
                          WRITE_SIZE_KB_mean=round(w_mean, 1), WRITE_SIZE_KB_max=round(w_max, 1),
                          hbm_read_bytes_per_launch_max=int(2 * f_max * 1024), hbm_write_bytes_per_launch_max=int(w_max * 1024),
                          hbm_bytes_per_launch_max=int((2 * f_max + w_max) * 1024))
        json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json"), "w"), indent=1)
        for k, v in out.items():
            print(k[:70], v["hbm_read_bytes_per_launch_max"] / 1e9, v["hbm_write_bytes_per_launch_max"] / 1e9)


if __name__ == "__main__":
    main()
