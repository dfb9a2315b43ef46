#!/bin/bash
# A/B of an environment knob of the PRODUCT library through bench.py, alternating runs on one box:
#   tools/experiments/ab_envp.sh <config> <VAR> "<v1> <v2> ..." [rounds] ['extra bench flags']
CFG=$1; VAR=$2; VALS=$3; R=${4:-2}; EXTRA=${5:-}
for r in $(seq $R); do
  for v in $VALS; do
    env $VAR=$v python bench.py --config $CFG $EXTRA --no-cpu --no-probes --steps 150 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$VAR=$v kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'ms/step', j['ms_per_step'], 'ber', j['config']['bit_error_rate_frame0'])"
  done
done
