#!/bin/bash
# aligned vs per-frame random leads (bench.py --lead ...), one box: sync_leads.sh [out]
out=${1:-gpurun_out/leads/sync_leads.txt}
mkdir -p "$(dirname "$out")"; : > "$out"
for c in cfg2 n1024 n4096 cfg3; do
  for l in aligned random; do
    python bench.py --config $c --lead $l --no-cpu --no-probes --steps 150 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-6s %-8s sync_ms %.4f demod_ms %.4f ms/step %.4f ber %.3g' % ('$c', '$l', r['sync_kernel_ms'], r['kernel_ms'], j['ms_per_step'], j['config']['bit_error_rate_frame0']))" >> "$out" || exit 1
  done
done
cat "$out"
