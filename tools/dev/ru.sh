#!/bin/bash
# tools/dev/ru.sh <filter-regex> <file.hip> [flags]: compact resource-usage lines
flt=$1; shift
"$(dirname "$0")/resuse.sh" "$@" 2>&1 | python3 -c "
import sys,ast,re
for l in sys.stdin:
    try: d=ast.literal_eval(l)
    except Exception: continue
    if re.search(r'''$flt''', d['name']): print(d['name'][20:75], 'VGPR',d['VGPRs'],'spill',d['VGPRs Spill'],'scratch',d['ScratchSize [bytes/lane]'],'occ',d['Occupancy [waves/SIMD]'],'SGPR',d['TotalSGPRs'])"
