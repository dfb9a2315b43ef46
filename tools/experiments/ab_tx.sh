#!/bin/bash
# tx_modulate rate, product library against a study library, alternating on one box: ab_tx.sh <study.so> [out]
lib=$1; out=${2:-gpurun_out/abtx/ab_tx.txt}
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2; do
  for cfg in "2048 144 1200 16QAM 4369" "2048 144 1200 64QAM 4369" "4096 288 2400 16QAM 2184" "1024 72 600 16QAM 8738"; do
    echo "## product   $cfg" >> "$out"; python tools/tx_rate.py $cfg 2>&1 | grep tx_modulate >> "$out" || exit 1
    echo "## $(basename $lib) $cfg" >> "$out"; OFDM_MI355X_LIB=$lib python tools/tx_rate.py $cfg 2>&1 | grep tx_modulate >> "$out" || exit 1
  done
done
cat "$out"
