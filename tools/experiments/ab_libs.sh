#!/bin/bash
# A/B of whole library builds through bench.py itself, alternating runs on one box:
#   tools/experiments/ab_libs.sh <config> "<tagA> <tagB> ..." [rounds] ['extra bench flags']
# tag "prod" = the product library; any other tag = tools/experiments/libofdm_g_<tag>.so (make geom TAG=<tag> GEOMFLAGS=...)
CFG=$1; TAGS=$2; R=${3:-3}; EXTRA=${4:-}
for r in $(seq $R); do
  for t in $TAGS; do
    if [ "$t" = prod ]; then unset OFDM_MI355X_LIB; else export OFDM_MI355X_LIB=$PWD/tools/experiments/libofdm_g_$t.so; fi
    python bench.py --config $CFG $EXTRA --no-cpu --no-probes --steps 150 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$t kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'ms/step', j['ms_per_step'], 'median', j['ms_per_step_median'], 'ber', j['config']['bit_error_rate_frame0'])"
  done
done
