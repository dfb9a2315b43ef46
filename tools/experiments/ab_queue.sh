#!/bin/bash
# work queue on / off per config, alternating on one box: ab_queue.sh [out]
out=${1:-gpurun_out/abq/ab_queue.txt}
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2; do
  for c in n1024 cfg2 n4096 cfg3; do
    for q in 1 0; do
      OFDM_MI355X_DEMOD_QUEUE=$q python bench.py --config $c --no-cpu --no-probes --steps 100 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$c queue=$q kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'ms/step', j['ms_per_step'])" >> "$out" || exit 1
    done
  done
done
cat "$out"
