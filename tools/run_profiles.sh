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
  # kernel trace: bench.py's DEFAULT step counts (250 timed + 10 warm-up), so that the mean in the stats table is the sustained
  # figure the bench line reports and not dominated by the first launches after an idle GPU (5.4, 5.3, 5.0, 4.7 ms ... at cfg2)
  rm -rf $OUT/kt
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --config $CFG --no-cpu --no-probes > $OUT/kt.json 2> $OUT/kt.err)
  # the plain bench line of the same build (no profiler attached): what profiles/<tag>_bench_<config>.json is made from
  python3 $R/bench.py --config $CFG --no-cpu > $OUT/bench.json 2> $OUT/bench.err
  if [ "${ONLY_KT:-0}" = "1" ]; then echo "$CFG: $(tail -c 300 $OUT/kt.json | head -c 300)"; continue; fi
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err)
  if [ "${PROFILE_SQ:-0}" = "1" ]; then
    (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/sq1 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/sq1.err)
    (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/sq2.err)
    (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq3 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/sq3.err)
  fi
  echo "$CFG: $(tail -c 300 $OUT/kt.json | head -c 300)"
done
# optional SQ counter passes (instruction mix, LDS bank conflicts, wave cycles) for the FIRST config: PROFILE_SQ=1 tools/run_profiles.sh ...
