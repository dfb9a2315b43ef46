#!/bin/bash
# On the GPU box: wall-clock rates of the kernels that are not on the metric's path (tx_modulate, channel, stand-alone de-mapper),
# then rocprofv3 kernel-trace + FETCH_SIZE / WRITE_SIZE passes of the same tools (program directly after `--`).
#   tools/kernel_rates.sh <tag>      -> gpurun_out/<tag>/{tx_rate.txt,demap_rate.txt,kt_tx,kt_demap,fetch_*,write_*}
set -u
TAG=$1; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python3 tools/tx_rate.py 2048 144 1200 16QAM 512 > $OUT/tx_rate.txt 2>&1
python3 tools/tx_rate.py 2048 144 1200 64QAM 512 >> $OUT/tx_rate.txt 2>&1
python3 tools/tx_rate.py 1024 72 600 16QAM 1024 >> $OUT/tx_rate.txt 2>&1
python3 tools/tx_rate.py 4096 288 2400 16QAM 256 >> $OUT/tx_rate.txt 2>&1
# the same at the bench's own batch sizes (the 512-frame launches above are the round-2 protocol, kept for comparison)
python3 tools/tx_rate.py 2048 144 1200 16QAM 4369 >> $OUT/tx_rate.txt 2>&1
python3 tools/tx_rate.py 2048 144 1200 64QAM 4369 >> $OUT/tx_rate.txt 2>&1
python3 tools/tx_rate.py 2048 144 1200 QPSK 4369 >> $OUT/tx_rate.txt 2>&1
python3 tools/tx_rate.py 1024 72 600 16QAM 8738 >> $OUT/tx_rate.txt 2>&1
python3 tools/tx_rate.py 4096 288 2400 16QAM 2184 >> $OUT/tx_rate.txt 2>&1
python3 tools/demap_rate.py > $OUT/demap_rate.txt 2>&1
python3 tools/stream_rate.py > $OUT/stream_rate.txt 2>&1
python3 tools/host_copy_probe.py >> $OUT/stream_rate.txt 2>&1
if [ "${RATES_ONLY:-0}" = "1" ]; then exit 0; fi
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_tx -- python3 $R/tools/tx_rate.py 2048 144 1200 16QAM 512 > $OUT/kt_tx.txt 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_demap -- python3 $R/tools/demap_rate.py > $OUT/kt_demap.txt 2>&1)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_tx -- python3 $R/tools/tx_rate.py 2048 144 1200 16QAM 512 > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_tx -- python3 $R/tools/tx_rate.py 2048 144 1200 16QAM 512 > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_demap -- python3 $R/tools/demap_rate.py > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_demap -- python3 $R/tools/demap_rate.py > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/sq_tx -- python3 $R/tools/tx_rate.py 2048 144 1200 16QAM 512 > /dev/null 2>&1)
echo done
