#!/bin/bash
# like ab_env.sh with a list of "VAR=val,VAR2=val2" settings: tools/experiments/ab_env2.sh <config> <lib tag> "<set1> <set2> ..." [rounds]
CFG=$1; TAG=$2; SETS=$3; R=${4:-2}
export OFDM_MI355X_LIB=$PWD/tools/experiments/libofdm_g_$TAG.so
for r in $(seq $R); do
  for st in $SETS; do
    env $(echo $st | tr ',' ' ') python bench.py --config $CFG --no-cpu --no-probes --steps 150 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$st kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'ms/step', j['ms_per_step'])"
  done
done
