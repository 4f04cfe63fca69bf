#!/bin/bash
# Gradient-mode traces only (one target and N in lockstep): per-kernel totals and the ordered launch list of the last step.
#   gpurun --timeout 600 -- 'bash tools/grad_trace_pass.sh gpurun_out/g1 8'
set -e
D=$1; LS=${2:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$D"
rocprofv3 --kernel-trace -d $D/gtrace1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --gradient-steps 6 --gradient-lockstep 0 --targets 0 --landmark-callback none --objectives 0 > $D/g1.json 2> $D/g1.err
DB1=$(ls -t $(find $D/gtrace1 -name "*_results.db") | head -1)
python3 tools/grad_step_trace.py $DB1 70 > $D/g1_totals.txt
python3 tools/grad_step_trace.py $DB1 --ordered > $D/g1_ordered.txt
if [ "$LS" != "0" ]; then
rocprofv3 --kernel-trace -d $D/gtraceN -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --gradient-steps 6 --gradient-lockstep $LS --targets 0 --landmark-callback none --objectives 0 > $D/gN.json 2> $D/gN.err
DBN=$(ls -t $(find $D/gtraceN -name "*_results.db") | head -1)
python3 tools/grad_step_trace.py $DBN 70 > $D/gN_totals.txt
python3 tools/grad_step_trace.py $DBN --ordered > $D/gN_ordered.txt
fi
rm -rf $D/gtrace1 $D/gtraceN
echo done
