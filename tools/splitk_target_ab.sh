#!/bin/bash
# Split-K fan-out of the tap-list kernel at one target (gradient mode): tools/build_exp.sh skt "" conv_taps.hip, then on the GPU box
#   bash tools/splitk_target_ab.sh OUTDIR     (MGF_SPLITK_TARGET = workgroups wanted, MGF_SPLITK_BELOW = used below this many)
out=${1:-gpurun_out/skt}; mkdir -p $out
export MGF_LIB_PATH=$PWD/exp_build/libmgf_skt.so
for cfg in "1024 256" "512 256" "768 256" "256 256" "512 128" "1024 256"; do
  set -- $cfg
  MGF_SPLITK_TARGET=$1 MGF_SPLITK_BELOW=$2 python bench.py --steps 2 --warmup 1 --gradient-steps 30 --gradient-lockstep 0 --no-cpu-baseline 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['gradient_mode']; print('target $1 below $2:', d['value'], 'iters/s', d['ms_per_step'], 'ms')" || exit 1
done | tee $out/ab.txt
