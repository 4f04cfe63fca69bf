#!/bin/bash
# One measurement pass on the GPU box (inside ONE gpurun call, i.e. on one MI355X): the default bench line, a kernel trace of the same
# command, the three counter passes, two gradient-mode traces; reduced on the box into OUT/profiles (raw rocprofv3 output stays there).
#   gpurun --timeout 1100 -- 'bash tools/measure_pass.sh gpurun_out/r3m 32 r3'
set -e
D=$1; B=${2:-32}; R=${3:-r3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$D"; mkdir -p "$D"
echo "[measure] bench"; python3 bench.py > $D/bench.json 2> $D/bench.err
echo "[measure] kernel trace"; rocprofv3 --kernel-trace --stats -d $D/trace -- python3 bench.py --steps 100 --warmup $B --no-cpu-baseline --gradient-steps 0 --targets 0 --landmark-callback none > $D/trace_bench.json 2> $D/trace.err
ARGS="--steps $B --warmup $B --batch $B --no-graph --no-cpu-baseline --gradient-steps 0 --targets 0 --landmark-callback none"
echo "[measure] pmc fetch"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 bench.py $ARGS > $D/pmc_fetch.json 2> $D/pmc_fetch.err
echo "[measure] pmc write"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 bench.py $ARGS > $D/pmc_write.json 2> $D/pmc_write.err
echo "[measure] pmc mfma"; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $D/pmc_mfma -- python3 bench.py $ARGS > $D/pmc_mfma.json 2> $D/pmc_mfma.err
echo "[measure] gradient traces"
rocprofv3 --kernel-trace -d $D/gtrace1 -- python3 bench.py --steps $B --warmup $B --no-cpu-baseline --gradient-steps 6 --gradient-lockstep 0 --targets 0 --landmark-callback none > $D/g1.json 2> $D/g1.err
rocprofv3 --kernel-trace -d $D/gtrace8 -- python3 bench.py --steps $B --warmup $B --no-cpu-baseline --gradient-steps 6 --gradient-lockstep 8 --targets 0 --landmark-callback none > $D/g8.json 2> $D/g8.err
echo "[measure] reduce"
bash tools/refresh_profiles.sh $D $B $R
{ echo "== one target (n = 1), the last step of the gradient-mode leg: python tools/grad_step_trace.py <rocprofv3 --kernel-trace db>";
  python3 tools/grad_step_trace.py $(ls -t $(find $D/gtrace1 -name "*_results.db") | head -1) 60;
  echo; echo "== 8 targets in lockstep (n = 8)";
  python3 tools/grad_step_trace.py $(ls -t $(find $D/gtrace8 -name "*_results.db") | head -1) 60; } > profiles/${R}_gradient_step_trace.txt
mkdir -p $D/profiles && cp profiles/${R}_* $D/profiles/
rm -rf $D/trace $D/pmc_fetch $D/pmc_write $D/pmc_mfma $D/gtrace1 $D/gtrace8          # raw output: too large to travel back
echo "[measure] done"; ls -la $D/profiles
