X="--bf16x3-leg 0 --no-cpu-baseline --gradient-steps 0 --targets 0 --objectives 0 --landmark-callback none --config4 0 --config5-targets 0 --arith bf16x3"
mkdir -p gpurun_out/bfx
for r in 128 256 512; do
  MGF_BF_DIRECT_MAX_RES=$r python bench.py $X > gpurun_out/bfx/r$r.json 2> gpurun_out/bfx/r$r.err && python -c "
import json,sys
d=json.loads(open('gpurun_out/bfx/r$r.json').read().strip().splitlines()[-1]); print('BFX max_res $r:', d['value'], 'iters/s', d['ms_per_step'], 'ms', d['hbm_gib'],'GiB')" || { tail -5 gpurun_out/bfx/r$r.err; exit 1; }
done
