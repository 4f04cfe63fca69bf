#!/bin/bash
# Build an experiment variant of the library: tools/build_exp.sh NAME "-DMGF_EXP=3 ..." [source.hip]  -> exp_build/libmgf_NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p exp_build/_obj_$1
C=morphganformer_amd/csrc
for s in capi.cpp bias_act.hip upfirdn2d.hip latent_prep.hip attention.hip losses.hip lpips_stem.hip embed.hip backward.hip conv_taps.hip wino.hip wino3.hip pointwise.hip narrow_conv.hip warp.hip; do
  cp -u $C/_obj/$s.o exp_build/_obj_$1/$s.o
done
SRC=${3:-conv_taps.hip}      # third argument: the source the flags apply to
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=on -Wno-unused-result $2 -x hip -c $C/$SRC -o exp_build/_obj_$1/$SRC.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp_build/libmgf_$1.so exp_build/_obj_$1/*.o
echo exp_build/libmgf_$1.so
