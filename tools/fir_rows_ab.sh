#!/bin/bash
# Segment height of the streaming blur (rows a wave walks: -DMGF_FS_ROWS=16 / 32 / 64 / 128 experiment builds fsrN), 32 samples
D=${1:-gpurun_out/fsr}; mkdir -p $D
export MGF_MICRO_N=32
for v in ${FSR_LIST:-64 16 32 128 64}; do
  echo "== MGF_FS_ROWS=$v" | tee -a $D/ab.txt
  MGF_LIB_PATH=$PWD/exp_build/libmgf_fsr$v.so python tools/fir_micro.py 256 512 1024 2>$D/err.txt | cut -c1-60 | tee -a $D/ab.txt
done
