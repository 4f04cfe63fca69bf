set -e
mkdir -p gpurun_out/pwc
python -m pytest tests/test_hip_ops.py tests/test_hip_embed.py tests/test_hip_fuzz.py tests/test_hip_projection.py -x -q -m gpu > gpurun_out/pwc/tests.log 2>&1 || { tail -30 gpurun_out/pwc/tests.log; exit 1; }
tail -2 gpurun_out/pwc/tests.log
python bench.py --no-cpu-baseline --gradient-steps 0 --targets 0 --landmark-callback none --config4 0 --config5-targets 0 --bf16x3-leg 0 > gpurun_out/pwc/bench.json 2> gpurun_out/pwc/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/pwc/bench.json').read().strip().splitlines()[-1])
print('HEADLINE', d['value'], d['ms_per_step'], 'objectives', {k:v.get('value') for k,v in d.get('objectives',{}).items()} if isinstance(d.get('objectives'),dict) else d.get('objectives'))
PY
