#!/bin/bash
# usage: tools/dev/resuse.sh csrc-file.hip [extra hipcc flags]
# one line per kernel: name, VGPRs, spills, scratch, occupancy, LDS (hipcc -Rpass-analysis=kernel-resource-usage)
f=$1; shift
cd "$(dirname "$0")/../../lte-gnu-radio-code_amd/csrc" || exit 1
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/resuse.$$.o 2>&1 | python3 -c "
import sys,re
cur={}
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name') or t.startswith('Name'):
        if cur: print(cur)
        cur={'name':t.split(':',1)[1].strip()[:100]}
    else:
        k,_,v=t.partition(':'); cur[k.strip()]=v.strip()
if cur: print(cur)
"
rm -f /tmp/resuse.$$.o
