#!/bin/bash
# 128-lane tiles for the transposed conv (tools/patches/tconv_stagger_wn.patch, tools/build_exp.sh stg "" conv_taps.hip)
D=gpurun_out/twn; mkdir -p $D
export MGF_MICRO_N=32 MGF_LIB_PATH=$PWD/exp_build/libmgf_stg.so
for v in "0 2" "1 2" "1 3" "0 2"; do set -- $v
  echo "== MGF_TCONV_WN=$1 MGF_RESIDENT=$2" | tee -a $D/ab.txt
  MGF_TCONV_WN=$1 MGF_RESIDENT=$2 python tools/conv_micro.py r128_tconv r256_tconv r512_tconv r1024_tconv 2>>$D/err.txt | tee -a $D/ab.txt
done
