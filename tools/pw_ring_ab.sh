#!/bin/bash
# same-box A/B of pointwise.hip ring variants: bash tools/pw_ring_ab.sh OUT  (exp_build/libmgf_pw_k*r*.so built by tools/build_exp.sh)
D=${1:-gpurun_out/pwab}; mkdir -p $D
for so in exp_build/libmgf_pw_k*.so; do
  tag=$(basename $so .so)
  MGF_LIB_PATH=$PWD/$so python tools/pw_ring_micro.py 16 > $D/$tag.txt 2> $D/$tag.err || { tail -5 $D/$tag.err; exit 1; }
  tail -1 $D/$tag.txt
done
paste <(cut -c1-28 $D/libmgf_pw_k4r2.txt) <(for f in $D/libmgf_pw_k*.txt; do :; done; for f in $D/libmgf_pw_k4r2.txt $D/libmgf_pw_k2r4.txt $D/libmgf_pw_k4r3.txt $D/libmgf_pw_k2r3.txt $D/libmgf_pw_k1r8.txt $D/libmgf_pw_k2r6.txt; do cut -c29-40 $f > $f.col; done; paste $D/libmgf_pw_k4r2.txt.col $D/libmgf_pw_k2r4.txt.col $D/libmgf_pw_k4r3.txt.col $D/libmgf_pw_k2r3.txt.col $D/libmgf_pw_k1r8.txt.col $D/libmgf_pw_k2r6.txt.col) | head -14
