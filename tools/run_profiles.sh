#!/bin/bash
# On the GPU box: rocprofv3 passes of bench.py for the given configs (kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in
# passes of their own, as MI355X_MICROARCH.md prescribes).  Output under gpurun_out/<tag>/<config>/{kt,fetch,write}.
#   tools/run_profiles.sh r02 cfg2 cfg3 n1024 n4096
set -u
TAG=$1; shift
R=$PWD
export TMPDIR=/tmp
for CFG in "$@"; do
  OUT=$R/gpurun_out/$TAG/$CFG
  mkdir -p $OUT
  ARGS="--config $CFG --steps 20 --warmup 3 --no-cpu --no-probes"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $ARGS > $OUT/kt.json 2> $OUT/kt.err)
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err)
  echo "$CFG: $(tail -c 300 $OUT/kt.json | head -c 300)"
done
