#!/bin/bash
# A/B of an experiment-library kernel variant through bench.py itself, alternating runs on one box:
#   tools/experiments/ab_variant.sh <config> <variant> [rounds] ['extra bench flags']      (variant 0 = the shipped kernel)
CFG=$1; V=$2; R=${3:-3}; EXTRA=${4:-}
export OFDM_MI355X_LIB=$PWD/tools/experiments/libofdm_mi355x_exp.so
for r in $(seq $R); do
  for v in 0 $V; do
    OFDM_EXP_VARIANT=$v python bench.py --config $CFG $EXTRA --no-cpu --no-probes --steps 150 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('variant $v kernel_ms', j['roofline']['kernel_ms'], 'ms/step', j['ms_per_step'])"
  done
done
