#!/bin/bash
# Fewest form-3 Winograd workgroups that still beat the split-K tap-list launch (conv.WINOGRAD3_MIN_WGS), gradient mode at one target:  bash tools/wino3_min_wgs_ab.sh OUT
D=${1:-gpurun_out/w3min}; mkdir -p $D
for v in ${W3MIN_LIST:-512 256 128 64 512}; do
  MGF_WINOGRAD3_MIN_WGS=$v python bench.py --steps 2 --warmup 1 --gradient-steps 30 --gradient-lockstep 0 --no-cpu-baseline --bf16x3-leg 0 --targets 0 --objectives 0 --landmark-callback none --config4 0 --config5-targets 0 2>>$D/err.txt \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['gradient_mode']; print('min_wgs $v:', d['value'], 'iters/s', d['ms_per_step'], 'ms', d['launches_per_step'], 'launches')" || exit 1
done | tee $D/ab.txt
