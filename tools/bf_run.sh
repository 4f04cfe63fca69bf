#!/bin/bash
# float32 / bf16x3 / bf16x3 with the direct kernel up to 512^2: three bench.py passes on one box -> gpurun_out/r5i
mkdir -p gpurun_out/r5i
X="--bf16x3-leg 0 --no-cpu-baseline --gradient-steps 0 --targets 0 --objectives 0 --landmark-callback none --config4 0 --config5-targets 0"
python bench.py $X > gpurun_out/r5i/f32.json 2>gpurun_out/r5i/f32.err
python bench.py $X --arith bf16x3 > gpurun_out/r5i/bf.json 2>gpurun_out/r5i/bf.err
MGF_BF_DIRECT_MAX_RES=512 python bench.py $X --arith bf16x3 > gpurun_out/r5i/bf512.json 2>gpurun_out/r5i/bf512.err
python - <<'PY'
import json
for t in ("f32","bf","bf512"):
    try:
        d=json.loads(open(f"gpurun_out/r5i/{t}.json").read().strip().splitlines()[-1])
        print("ARITH",t,d["value"],d["ms_per_step"],d["roofline"]["kernel"],d["roofline"]["avg_launch_us"], {k:(v["launches_per_iter"],v["avg_us"]) for k,v in d["roofline"]["all_conv_kernels"].items()})
    except Exception as e: print("ARITH",t,"failed",e, open(f"gpurun_out/r5i/{t}.err").read()[-1500:])
PY
