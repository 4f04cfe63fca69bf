#!/bin/bash
# One lean headline run (and, with G=1, one gradient-mode run at one target) per setting of the library's dispatch knobs; library with tuning hooks in EVERY source:
#   tools/build_exp.sh hooks "" all;   bash tools/knob_sweep.sh OUT "MGF_PW_SPLITK_WAVES=0 MGF_PW_SPLITK_WAVES=1024 ..."      (same box; "base" = no knob)
D=${1:-gpurun_out/knobs}; mkdir -p $D
export MGF_LIB_PATH=$PWD/exp_build/libmgf_hooks.so
X="--bf16x3-leg 0 --no-cpu-baseline --targets 0 --objectives 0 --landmark-callback none --config4 0 --config5-targets 0"
for kv in base $2 base; do
  if [ "$kv" = base ]; then E="MGF_NOOP=1"; else E="$kv"; fi
  if [ "${G:-0}" = 1 ]; then
    env $E python bench.py $X --steps 2 --warmup 1 --gradient-steps 30 --gradient-lockstep 0 2>>$D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['gradient_mode']; print('$kv gradient', d['value'], d['ms_per_step'])" || exit 1
  else
    env $E python bench.py $X --gradient-steps 0 2>>$D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv headline', d['value'], d['ms_per_step'])" || exit 1
  fi
done | tee -a $D/sweep.txt
