#!/bin/bash
# Start-phase spread of the transposed conv's persistent workgroups (tools/build_exp.sh stg "" conv_taps.hip): MGF_TC_STAGGER = N/16 chunk periods, -N = N/16 tile periods
D=${1:-gpurun_out/stg}; mkdir -p $D
export MGF_MICRO_N=32 MGF_LIB_PATH=$PWD/exp_build/libmgf_stg.so
for v in 0 8 16 32 64 128 -4 -8 -16 0; do
  echo "== MGF_TC_STAGGER=$v" | tee -a $D/ab.txt
  MGF_TC_STAGGER=$v python tools/conv_micro.py r128_tconv r256_tconv r512_tconv r1024_tconv 2>$D/err.txt | tee -a $D/ab.txt
done
