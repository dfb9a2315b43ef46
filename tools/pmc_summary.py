#!/usr/bin/env python3
"""Print per-kernel means of every counter found in rocprofv3 counter_collection CSVs under the given dirs."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    import os
    for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if "ofdm::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-28s %16.1f  (n=%d)" % (n, sum(v) / len(v), len(v)))
