#!/bin/bash
# Timing ablations of the persistent form-3 Winograd kernel (experiment builds: tools/build_exp.sh w3ablK "-DW3P_ABL=K" wino3.hip), same box:
#   bash tools/w3p_abl.sh OUT     (1 no output stores, 2 no epilogue, 3 no input traffic, 4 no matrix work, 5 no epilogue-operand DMA, 6 = 1 + 3)
D=${1:-gpurun_out/w3p_abl}; mkdir -p $D
for v in w3base w3abl1 w3abl2 w3abl3 w3abl4 w3abl5 w3abl6 w3base; do
  MGF_LIB_PATH=$PWD/exp_build/libmgf_$v.so python tools/w3_top_micro.py 2>$D/$v.err | tee -a $D/abl.txt
done
