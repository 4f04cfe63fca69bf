#!/bin/bash
# headline variants, one bench.py process each (no extra legs):  bash tools/headline_ab.sh OUTDIR
D=${1:-gpurun_out/hab}; mkdir -p $D
run() { tag=$1; shift; env "$@" > $D/$tag.json 2>> $D/err.log; python - "$D/$tag.json" "$tag" <<'PY' | tee -a $D/ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"HEADLINE {sys.argv[2]}: {d['value']} iters/s, {d['ms_per_step']} ms/step, hbm {d['hbm_gib']} GiB, frac {d['roofline']['frac']} ({d['roofline']['kernel']})")
except Exception as e:
    print("HEADLINE", sys.argv[2], "failed", e)
PY
}
X="--bf16x3-leg 0 --no-cpu-baseline --gradient-steps 0 --targets 0 --objectives 0 --landmark-callback none --config4 0 --config5-targets 0"
run b32      MGF_D=0 python bench.py $X
run b48      MGF_D=0 python bench.py $X --batch 48 --steps 14
run b64      MGF_D=0 python bench.py $X --batch 64 --steps 10
run pipe     MGF_D=0 python bench.py $X --pipeline 1
run pipe_ovl MGF_OVERLAP_SKIP=1 python bench.py $X --pipeline 1
run ovl      MGF_OVERLAP_SKIP=1 python bench.py $X
