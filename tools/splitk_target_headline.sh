#!/bin/bash
# Split-K fan-out of the tap-list kernel, headline and objective legs (tools/build_exp.sh skt "" conv_taps.hip first): bash tools/splitk_target_headline.sh
D=gpurun_out/skth; mkdir -p $D
export MGF_LIB_PATH=$PWD/exp_build/libmgf_skt.so
X="--bf16x3-leg 0 --no-cpu-baseline --gradient-steps 0 --targets 0 --objectives 0 --landmark-callback none --config4 0 --config5-targets 0"
for t in 1024 512 1024 512; do
  MGF_SPLITK_TARGET=$t python bench.py $X 2>>$D/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('target $t headline', d['value'], d['ms_per_step'])" || exit 1
done | tee $D/ab.txt
for t in 1024 512; do
  MGF_SPLITK_TARGET=$t python bench.py --steps 4 --warmup 1 --no-cpu-baseline --bf16x3-leg 0 --gradient-steps 0 --targets 0 --landmark-callback none --config4 0 --config5-targets 0 2>>$D/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['objectives']; print('target $t', {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.items()})" || exit 1
done | tee -a $D/ab.txt
