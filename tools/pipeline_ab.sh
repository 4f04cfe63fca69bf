#!/bin/bash
# Headline with and without the two-stream pipeline (ProjectionEngine(pipeline=True)), alternated on one box
X="--bf16x3-leg 0 --no-cpu-baseline --gradient-steps 0 --targets 0 --objectives 0 --landmark-callback none --config4 0 --config5-targets 0"
for i in 1 2 3 4; do for p in 0 1; do
python bench.py $X --pipeline $p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $p:', d['value'], d['ms_per_step'])"
done; done
