#!/bin/bash
# Demod rate of the small FFT sizes (64..512-pt), work queue on and off.  usage: small_sizes.sh [out]
out=${1:-gpurun_out/small/small_sizes.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
for c in cfgA n128 n256 n512; do
  for q in 1 0; do
    OFDM_MI355X_DEMOD_QUEUE=$q python bench.py --config $c --no-cpu --no-probes --steps 40 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$c queue=$q', 'Msamples/s', round(j['value']), 'kernel_ms', r['kernel_ms'], 'sync_ms', r['sync_kernel_ms'], 'frac', r['frac'], 'ber', j['config']['bit_error_rate_frame0'])" >> "$out" || exit 1
  done
done
cat "$out"
