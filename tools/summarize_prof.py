#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/<tag>/<config>/ (tools/run_profiles.sh) into the summaries committed under profiles/.

  tools/summarize_prof.py TAG CONFIG [CONFIG ...]

Writes profiles/TAG_<config>_kernel_stats.csv (rocprofv3's --stats table) and profiles/TAG_demod_traffic.json: per config the
demod kernel's HBM bytes per launch (mean over its launches) from the separate FETCH_SIZE / WRITE_SIZE passes, corrected as
MI355X_MICROARCH.md (HBM) prescribes -- FETCH_SIZE counts 1/2 of a coalesced stream read on gfx950, WRITE_SIZE is exact -- with
the correction re-validated in the same run on channel_kernel, whose byte count is known exactly (8 B/lane coalesced read and
write of every sample).  Each entry carries the kernel's name, the sha of the kernel sources it was measured on and the number
of data symbols per launch: bench.py quotes the figure only for a build and workload that match.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def pmc(dirname, counter):
    files = sorted(glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    agg = collections.defaultdict(list)
    for f in files:                                  # the newest run only (gpurun merges successive runs into the same tree)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, cfgs = sys.argv[1], sys.argv[2:]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    sha = bench.kernel_source_sha()
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    out = {}
    for cfg in cfgs:
        base = os.path.join(ROOT, "gpurun_out", tag, cfg)
        ks = sorted(glob.glob(os.path.join(base, "kt", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
        shutil.copy(ks, os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, cfg)))
        line = json.loads(open(os.path.join(base, "kt.json")).read().strip().splitlines()[-1])
        # the plain bench line is committed only if it was produced by the sources that are in the tree now
        bpath = os.path.join(base, "bench.json")
        if os.path.exists(bpath):
            bl = json.loads(open(bpath).read().strip().splitlines()[-1])
            got = bl["roofline"].get("kernel_source_sha")
            if got != sha:
                raise SystemExit("%s: bench line of kernel sources %s, the tree is %s -- re-run tools/run_profiles.sh" % (bpath, got, sha))
            with open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, cfg)), "w") as f:
                f.write(json.dumps(bl) + "\n")
        if line["roofline"].get("kernel_source_sha") != sha:
            raise SystemExit("%s/kt.json was measured on kernel sources %s, the tree is %s" % (base, line["roofline"].get("kernel_source_sha"), sha))
        fe, wr = pmc(os.path.join(base, "fetch"), "FETCH_SIZE"), pmc(os.path.join(base, "write"), "WRITE_SIZE")
        name = [k for k in fe if "rx_demod_kernel" in k][0]
        c = bench.CONFIGS[cfg]
        dsym = c["frames"] * (c["n_sym"] // 4) * 3
        # calibration on channel_kernel: reads and writes frames x frame_len x 8 B per launch (1 tap), 256 frames per launch
        chan = [k for k in fe if "channel_kernel" in k]
        calib = None
        if chan:
            fl = c["n_sym"] * (c["nfft"] + c["cp"])
            exact = max(256 * fl * 8, 1)
            calib = dict(exact_bytes_read_per_full_launch=exact, FETCH_SIZE_KB_max=max(fe[chan[0]]),
                         fetch_bytes_over_exact=round(max(fe[chan[0]]) * 1024 / exact, 4),
                         WRITE_SIZE_KB_max=max(wr[chan[0]]), write_bytes_over_exact=round(max(wr[chan[0]]) * 1024 / exact, 4))
        f_mean = sum(fe[name]) / len(fe[name])
        w_mean = sum(wr[name]) / len(wr[name])
        rd, wrb = 2.0 * f_mean * 1024, w_mean * 1024
        alg = line["roofline"]["algorithmic_bytes_per_launch"]
        out[cfg] = dict(kernel=name, kernel_source_sha=sha, git=git, data_symbols_per_launch=dsym, launches=len(fe[name]),
                        FETCH_SIZE_KB_mean=round(f_mean, 1), WRITE_SIZE_KB_mean=round(w_mean, 1),
                        hbm_read_bytes_per_launch_mean=int(rd), hbm_write_bytes_per_launch_mean=int(wrb),
                        hbm_bytes_per_launch_mean=int(rd + wrb), algorithmic_bytes_per_launch=alg,
                        traffic_over_algorithmic=round((rd + wrb) / alg, 4), calibration_on_channel_kernel=calib,
                        kernel_ms_under_kernel_trace=line["roofline"]["kernel_ms"])
        print(cfg, "read %.2f GB write %.2f GB = %.3f x algorithmic" % (rd / 1e9, wrb / 1e9, (rd + wrb) / alg), calib)
    path = os.path.join(ROOT, "profiles", "%s_demod_traffic.json" % tag)
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(out)
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
