#!/bin/bash
# 4096-pt sync search with random leads: look-ahead of 3 / 5 (product) / 7 blocks, alternating libraries on one box.
# Build the study libraries first:  make -C lte-gnu-radio-code_amd/csrc geom TAG=mx3 GEOMFLAGS=-DOFDM_SCAN_MX_T256=3  (and mx7 / =7).
for rep in 1 2; do for l in lte-gnu-radio-code_amd/ofdm_mi355x/libofdm_mi355x.so tools/experiments/libofdm_g_mx3.so tools/experiments/libofdm_g_mx7.so; do
  OFDM_MI355X_LIB=$l python bench.py --config n4096 --lead random --no-cpu --no-probes --steps 100 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$l'.split('/')[-1], 'sync_ms %.4f demod_ms %.4f ms/step %.4f ber %.3g' % (r['sync_kernel_ms'], r['kernel_ms'], j['ms_per_step'], j['config']['bit_error_rate_frame0']))"; done; done
