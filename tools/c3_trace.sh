set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/c3t; rm -rf $D; mkdir -p $D
python -m pytest tests/test_hip_embed.py tests/test_hip_fuzz.py -x -q -m gpu > $D/tests.log 2>&1 || { tail -20 $D/tests.log; exit 1; }
tail -2 $D/tests.log
rocprofv3 --kernel-trace --stats -d $D/trace -- python3 bench.py --workload config3 --config3-targets 2 --config3-steps 64 > $D/c3_bench.json 2> $D/c3.err
python3 tools/rocpd_stats.py $(ls -t $(find $D/trace -name "*_results.db") | head -1) > $D/c3_stats.txt
head -24 $D/c3_stats.txt | cut -c1-170
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c3t/c3_bench.json').read().strip().splitlines()[-1]); print('C3', d['value'], d.get('ms_per_step'))
PY
rm -rf $D/trace
