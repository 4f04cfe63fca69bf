#!/bin/bash
# Timing ablations of the ONE-SHOT form-3 Winograd kernel on the generator's conv1 layers (experiment builds of tools/patches/w3_oneshot_ablation.patch:
# tools/build_exp.sh w3oK "-DW3O_ABL=K" wino3.hip; 1 no output transform / epilogue, 2 no matrix work, 3 no input loads, 4 no weight loads,
# 5 epilogue operands requested in the epilogue instead of the prologue, 6 = 3 + 4), same box:  bash tools/w3o_abl.sh OUT
D=${1:-gpurun_out/w3o_abl}; mkdir -p $D
for v in w3obase w3o1 w3o2 w3o3 w3o4 w3o5 w3o6 w3obase; do
  MGF_LIB_PATH=$PWD/exp_build/libmgf_$v.so python tools/w3_layers_micro.py 2>$D/$v.err | tee -a $D/abl.txt
done
