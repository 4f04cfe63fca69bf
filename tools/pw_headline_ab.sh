#!/bin/bash
# same-box A/B of the headline + config-3 legs across pointwise.hip ring variants (exp_build/libmgf_pw_*.so)
# The variants (profiles/r5_pw_ring_ab.txt) are PW_KU k-steps per load group x PW_NR groups in the register ring, built with
#   for v in "4 2" "2 4" "4 3" "2 3" "1 8" "2 6"; do set -- $v; tools/build_exp.sh pw_k$1r$2 "-DPW_KU=$1 -DPW_NR=$2" pointwise.hip; done
# (run it here when exp_build/ is empty; the product default is the PW_KU / PW_NR pair at the top of csrc/pointwise.hip)
D=${1:-gpurun_out/pwh}; mkdir -p $D
ls exp_build/libmgf_pw_k*.so >/dev/null 2>&1 || for v in "4 2" "1 8" "2 4"; do set -- $v; bash tools/build_exp.sh pw_k$1r$2 "-DPW_KU=$1 -DPW_NR=$2" pointwise.hip; done
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
