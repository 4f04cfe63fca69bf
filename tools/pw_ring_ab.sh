#!/bin/bash
# same-box A/B of pointwise.hip ring variants: bash tools/pw_ring_ab.sh OUT  (exp_build/libmgf_pw_k*r*.so built by tools/build_exp.sh)
# The variants (profiles/r5_pw_ring_ab.txt) are PW_KU k-steps per load group x PW_NR groups in the register ring, built with
#   for v in "4 2" "2 4" "4 3" "2 3" "1 8" "2 6"; do set -- $v; tools/build_exp.sh pw_k$1r$2 "-DPW_KU=$1 -DPW_NR=$2" pointwise.hip; done
# (run it here when exp_build/ is empty; the product default is the PW_KU / PW_NR pair at the top of csrc/pointwise.hip)
D=${1:-gpurun_out/pwab}; mkdir -p $D
ls exp_build/libmgf_pw_k*.so >/dev/null 2>&1 || for v in "4 2" "2 4" "4 3" "2 3" "1 8" "2 6"; do set -- $v; bash tools/build_exp.sh pw_k$1r$2 "-DPW_KU=$1 -DPW_NR=$2" pointwise.hip; done
for so in exp_build/libmgf_pw_k*.so; do
  tag=$(basename $so .so)
  MGF_LIB_PATH=$PWD/$so python tools/pw_ring_micro.py 16 > $D/$tag.txt 2> $D/$tag.err || { tail -5 $D/$tag.err; exit 1; }
  tail -1 $D/$tag.txt
done
paste <(cut -c1-28 $D/libmgf_pw_k4r2.txt) <(for f in $D/libmgf_pw_k*.txt; do :; done; for f in $D/libmgf_pw_k4r2.txt $D/libmgf_pw_k2r4.txt $D/libmgf_pw_k4r3.txt $D/libmgf_pw_k2r3.txt $D/libmgf_pw_k1r8.txt $D/libmgf_pw_k2r6.txt; do cut -c29-40 $f > $f.col; done; paste $D/libmgf_pw_k4r2.txt.col $D/libmgf_pw_k2r4.txt.col $D/libmgf_pw_k4r3.txt.col $D/libmgf_pw_k2r3.txt.col $D/libmgf_pw_k1r8.txt.col $D/libmgf_pw_k2r6.txt.col) | head -14
