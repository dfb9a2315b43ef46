#!/usr/bin/env python3
"""Timeline of one call from a rocprofv3 results database (kernel + memory-copy trace): trace_timeline.py db [copy_bytes] [skip]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
mark = sys.argv[2] if len(sys.argv) > 2 else None
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ev = []
for r in db.execute("select name,start,end,grid_x,workgroup_x,stream_id from kernels"):
    ev.append((r[1], r[2], "K %s grid=%d wg=%d stream=%s" % (r[0][:48], r[3] // max(r[4], 1), r[4], r[5])))
for r in db.execute("select name,start,end,size from memory_copies"):
    ev.append((r[1], r[2], "C %s %d" % (r[0].replace("MEMORY_COPY_", ""), r[3])))
ev.sort()
idx = [i for i, e in enumerate(ev) if e[2].startswith("C") and (mark is None or e[2].endswith(" " + mark))]
i0 = idx[min(skip, len(idx) - 2)]
i1 = idx[min(skip, len(idx) - 2) + 1]
t0 = ev[i0][0]
for e in ev[i0:i1 + 1]:
    print("%9.1f us  dur %7.1f us  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
