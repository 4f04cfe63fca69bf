#!/bin/bash
# same-box A/B of the headline + config-3 legs across pointwise.hip ring variants (exp_build/libmgf_pw_*.so)
D=${1:-gpurun_out/pwh}; mkdir -p $D
X="--no-cpu-baseline --gradient-steps 0 --targets 0 --landmark-callback none --config4 0 --config5-targets 0 --bf16x3-leg 0"
for rep in 1 2; do
for so in exp_build/libmgf_pw_k4r2.so exp_build/libmgf_pw_k1r8.so exp_build/libmgf_pw_k2r4.so; do
  tag=$(basename $so .so)
  MGF_LIB_PATH=$PWD/$so python bench.py $X > $D/$tag.$rep.json 2> $D/$tag.$rep.err || { tail -5 $D/$tag.$rep.err; exit 1; }
  python - "$D/$tag.$rep.json" "$tag.$rep" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('AB', sys.argv[2], d['value'], d['ms_per_step'], {k:v.get('value') for k,v in d.get('objectives',{}).items()})
PY
done; done
