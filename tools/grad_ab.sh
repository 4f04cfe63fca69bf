#!/bin/bash
# A/B runs of the gradient-mode knobs (Python-level switches), one grad_time.py process each:  bash tools/grad_ab.sh OUTDIR
D=${1:-gpurun_out/ab}; mkdir -p $D
run() { tag=$1; shift; env "$@" python tools/grad_time.py ${B:-1} ${STEPS:-40} $tag 2>>$D/err.log | grep GRADTIME | tee -a $D/ab.txt; }
run base_all_off MGF_LPIPS_MERGE_FIRE=0 MGF_FUSE_ACT_FIR=0
run merge_fire   MGF_FUSE_ACT_FIR=0
run defaults     MGF_DUMMY=0
run no_border    MGF_TCONV_BORDER=0
run skip_s256    MGF_GRAD_SKIP_STREAM=256
run skip_s64     MGF_GRAD_SKIP_STREAM=64
B=16 STEPS=12 run b16_no_fir MGF_FUSE_ACT_FIR=0
B=16 STEPS=12 run b16_defaults MGF_DUMMY=0
B=16 STEPS=12 run b16_skip256 MGF_GRAD_SKIP_STREAM=256
