#!/bin/bash
# Timing ablations of the transposed-conv kernel (conv_taps_kernel<1,2,1,true,10>) on the generator's up-sampling layers at 32 samples: experiment builds of
# tools/patches/tconv_ablation.patch (git apply it, then  for v in 0 1 2 4 8 16 32 64 6 14; do tools/build_exp.sh tc$v "-DTC_ABL=$v" conv_taps.hip; done):
# 1 no epilogue, 2 no global loads, 4 no LDS staging stores, 8 no chunk barriers, 16 no LDS operand reads, 32 no epilogue stores, 64 no matrix instructions
# (bits add).  Same box:  bash tools/tconv_abl.sh OUT
D=${1:-gpurun_out/tc_abl}; mkdir -p $D
export MGF_MICRO_N=32
for v in 0 1 32 2 4 6 8 14 16 64 0; do
  echo "== TC_ABL=$v" | tee -a $D/abl.txt
  MGF_LIB_PATH=$PWD/exp_build/libmgf_tc$v.so python tools/conv_micro.py r128_tconv r256_tconv r512_tconv r1024_tconv 2>$D/tc$v.err | tee -a $D/abl.txt
done
