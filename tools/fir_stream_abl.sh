#!/bin/bash
# Timing ablations of the streaming blur (fir_up1_stream<true>) at 32 samples: experiment builds of tools/patches/fir_stream_ablation.patch
# (for v in 0 1 2 3 4 8; do tools/build_exp.sh fs$v "-DFS_ABL=$v" upfirdn2d.hip; done): 1 no halo loads, 2 no noise loads, 4 no stores, 8 no main loads
D=${1:-gpurun_out/fs_abl}; mkdir -p $D
export MGF_MICRO_N=32
for v in 0 1 2 3 4 8 0; do
  echo "== FS_ABL=$v" | tee -a $D/abl.txt
  MGF_LIB_PATH=$PWD/exp_build/libmgf_fs$v.so python tools/fir_micro.py 256 512 1024 2>$D/err.txt | tee -a $D/abl.txt
done
