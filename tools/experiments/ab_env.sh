#!/bin/bash
# A/B of one environment knob of a study build through bench.py, alternating runs on one box:
#   tools/experiments/ab_env.sh <config> <lib tag> <VAR> "<v1> <v2> ..." [rounds]
CFG=$1; TAG=$2; VAR=$3; VALS=$4; R=${5:-2}
export OFDM_MI355X_LIB=$PWD/tools/experiments/libofdm_g_$TAG.so
for r in $(seq $R); do
  for v in $VALS; do
    env $VAR=$v python bench.py --config $CFG --no-cpu --no-probes --steps 150 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$VAR=$v kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'ms/step', j['ms_per_step'])"
  done
done
