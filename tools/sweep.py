#!/usr/bin/env python3
"""Mixed-numerology sweep of BASELINE.json configs[4] on one GPU: {1024,2048,4096}-pt x {QPSK,16-QAM,64-QAM}.
Runs bench.py once per combination (own process each), prints a markdown table and, with --json FILE, writes the nine result
lines (bench.py's own JSON, one per cell) to FILE."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
rows, lines = [], []
for cfg in ("n1024", "cfg2", "n4096"):
    for mod in ("QPSK", "16QAM", "64QAM"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--mod", mod, "--no-cpu", "--steps", "100", "--warmup", "5"],
                             capture_output=True, text=True, timeout=600)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        lines.append(d)
        r = d["roofline"]
        rows.append((d["config"]["workload"].split(",")[0], mod, d["value"] / 1e3, r["kernel_ms"], r["achieved"], r["frac"],
                     r["measured_copy_GBs"], r["package_power_w_and_sclk_mhz_under_load"], d["config"]["bit_error_rate_frame0"]))
        print(rows[-1], flush=True)
if out_json:
    with open(out_json, "w") as f:
        for d in lines:
            f.write(json.dumps(d) + "\n")
print("| numerology | constellation | Gsamples/s | demod kernel ms | algorithmic GB/s | frac of 8 TB/s | same-run copy GB/s | power W, sclk MHz | BER frame 0 |")
print("|---|---|---|---|---|---|---|---|---|")
for w, m, v, k, a, f, c, pw, b in rows:
    print("| %s | %s | %.0f | %.3f | %.0f | %.3f | %.0f | %s | %.1e |" % (w, m, v, k, a, f, c, pw, b))
