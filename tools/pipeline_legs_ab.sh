#!/bin/bash
# The bench legs with and without the two-stream pipeline, alternated on one box: objectives (LPIPS(vgg), config 3 at 16 / 32 candidates), many_targets, config 4 / 5
for i in 1 2; do for p in 0 1; do
python bench.py --pipeline $p --steps 4 --warmup 1 --bf16x3-leg 0 --no-cpu-baseline --gradient-steps 0 --landmark-callback none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pipeline $p:', {k: v.get('value') for k, v in d['objectives'].items()}, 'literal-1000', d['many_targets']['retargeted_repeats']['iters_per_s'], 'config4', d['config4']['value'], 'config5', d['config5']['value'])"
done; done
