#!/usr/bin/env python3
"""tools/dev/isa_blocks.py <file.hip> <kernel-name-regex> [hipcc flags]: per basic block instruction mix of one kernel
(pk = packed VALU, v = other VALU, ds = LDS, vm = global/scratch, s = SALU, br = branches), and the totals."""
import collections, os, re, subprocess, sys
src, pat, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "lte-gnu-radio-code_amd", "csrc")
out = "/tmp/isa_blocks.s"
subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-Wno-unused-function",
                "--cuda-device-only", "-S", src, "-o", out] + flags, cwd=csrc, check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
m = re.search(r"^(%s\w*):" % pat, s, re.M)
body = s[m.end():]
body = body[:body.index("s_endpgm")]
def cls(l):
    op = l.split()[0]
    for p, c in (("v_pk_", "pk"), ("v_", "v"), ("ds_", "ds"), ("global_", "vm"), ("scratch_", "vm"), ("s_waitcnt", "w"), ("s_barrier", "BAR"),
                 ("s_cbranch", "br"), ("s_branch", "br"), ("s_", "s")):
        if op.startswith(p):
            return c
    return "o"
blocks, cur = [], ["entry", []]
for l in body.split("\n"):
    l = l.strip()
    if not l or l.startswith(";"):
        continue
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        blocks.append(cur)
        cur = [mm.group(1) + (" LOOP" if "Loop Header" in l else ""), []]
        continue
    if l.startswith("."):
        continue
    cur[1].append(l)
blocks.append(cur)
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter(cls(l) for l in ins)
    tot.update(c)
    print("%-14s %4d  %s  | %s" % (name, len(ins), dict(c), ins[-1][:60] if ins else ""))
print("TOTAL", dict(tot))
