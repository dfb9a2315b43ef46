#!/usr/bin/env python3
"""Condense tools/kernel_rates.sh output (gpurun_out/<tag>/) into the summaries committed under profiles/:

  tools/summarize_rates.py TAG OUT_PREFIX      e.g.  tools/summarize_rates.py r03k r03

writes profiles/<OUT>_{tx,demap}_kernel_stats.csv (rocprofv3 --stats tables), <OUT>_{tx,demap,stream}_rate.txt (wall-clock prints),
<OUT>_tx_pmc_sq.txt (SQ counters) and <OUT>_other_kernels_trace_pmc.json: per kernel the kernel-trace mean duration and the HBM bytes per
launch from the separate FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md; WRITE_SIZE as counted; both are
reported in KB by rocprofv3), their sum over the mean duration and its fraction of 8 TB/s.  Only the newest run under each directory
is read (gpurun merges successive runs into one tree)."""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(d, pat):
    fs = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def counter(d, name):
    agg = collections.defaultdict(list)
    f = newest(d, "*_counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    tag, out = sys.argv[1], sys.argv[2]
    base, prof = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    rows = []
    for what in ("tx", "demap"):
        ks = newest(os.path.join(base, "kt_" + what), "*_kernel_stats.csv")
        shutil.copy(ks, os.path.join(prof, "%s_%s_kernel_stats.csv" % (out, what)))
        fetch, write = counter(os.path.join(base, "fetch_" + what), "FETCH_SIZE"), counter(os.path.join(base, "write_" + what), "WRITE_SIZE")
        for r in csv.DictReader(open(ks)):
            k = r["Name"]
            if "ofdm::" not in k:
                continue
            rd, wr = fetch.get(k, 0.0) * 1024 * 2, write.get(k, 0.0) * 1024
            us = float(r["AverageNs"]) / 1e3
            rows.append(dict(kernel=k, calls=int(r["Calls"]), avg_us=round(us, 1), hbm_read_bytes=int(rd), hbm_write_bytes=int(wr),
                             hbm_GBs=round((rd + wr) / us / 1e3, 1), frac_of_8TBs=round((rd + wr) / us / 1e3 / 8000, 3)))
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    with open(os.path.join(prof, out + "_other_kernels_trace_pmc.json"), "w") as f:
        json.dump(dict(source="tools/kernel_rates.sh " + tag + " (512-frame launches of tools/tx_rate.py, tools/demap_rate.py)", git=git, kernels=rows), f, indent=1)
    for name in ("tx_rate", "demap_rate", "stream_rate"):
        src = os.path.join(base, name + ".txt")
        if os.path.exists(src):
            txt = [l for l in open(src) if "amdgpu.ids" not in l]
            open(os.path.join(prof, "%s_%s.txt" % (out, name)), "w").writelines(txt)
    sq = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(base, "sq_tx")], capture_output=True, text=True).stdout
    open(os.path.join(prof, out + "_tx_pmc_sq.txt"), "w").write(sq)
    for r in rows:
        print("%-80s %8.1f us  %6.1f GB/s  %.3f" % (r["kernel"][:80], r["avg_us"], r["hbm_GBs"], r["frac_of_8TBs"]))


if __name__ == "__main__":
    main()
