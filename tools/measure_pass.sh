#!/bin/bash
# One measurement pass on the GPU box (inside ONE gpurun call, i.e. on one MI355X): the default bench line, a kernel trace of the same
# command, the three counter passes, two gradient-mode traces; reduced on the box into OUT/profiles (raw rocprofv3 output stays there).
#   gpurun --timeout 1100 -- 'bash tools/measure_pass.sh gpurun_out/r3m 32 r3'
set -e
D=$1; B=${2:-32}; R=${3:-r5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$D"; mkdir -p "$D"
echo "[measure] bench"; python3 bench.py > $D/bench.json 2> $D/bench.err
echo "[measure] kernel trace"; LEAN="--no-cpu-baseline --gradient-steps 0 --targets 0 --landmark-callback none --objectives 0 --config4 0 --config5-targets 0 --bf16x3-leg 0"
rocprofv3 --kernel-trace --stats -d $D/trace -- python3 bench.py --steps 20 --warmup 2 $LEAN > $D/trace_bench.json 2> $D/trace.err
# the same on ONE stream (--pipeline 0): kernels running alone, for the per-iteration analyses (iteration trace / floor) and the counter passes -- in the
# default two-stream schedule the loss phase's kernels run beside the generator's and their durations (and per-kernel busy counters) describe the pair
echo "[measure] kernel trace, one stream"
rocprofv3 --kernel-trace --stats -d $D/trace1 -- python3 bench.py --steps 20 --warmup 2 --pipeline 0 $LEAN > $D/trace1_bench.json 2> $D/trace1.err
ARGS="--steps 1 --warmup 1 --batch $B --no-graph --pipeline 0 $LEAN"
echo "[measure] pmc fetch"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 bench.py $ARGS > $D/pmc_fetch.json 2> $D/pmc_fetch.err
echo "[measure] pmc write"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 bench.py $ARGS > $D/pmc_write.json 2> $D/pmc_write.err
echo "[measure] pmc mfma"; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $D/pmc_mfma -- python3 bench.py $ARGS > $D/pmc_mfma.json 2> $D/pmc_mfma.err
echo "[measure] gradient traces"
GL="--no-cpu-baseline --targets 0 --landmark-callback none --objectives 0 --config4 0 --config5-targets 0 --bf16x3-leg 0"
rocprofv3 --kernel-trace -d $D/gtrace1 -- python3 bench.py --steps 1 --warmup 1 $GL --gradient-steps 6 --gradient-lockstep 0 > $D/g1.json 2> $D/g1.err
rocprofv3 --kernel-trace -d $D/gtrace8 -- python3 bench.py --steps 1 --warmup 1 $GL --gradient-steps 6 --gradient-lockstep 16 > $D/g8.json 2> $D/g8.err
echo "[measure] bf16x3 mode: kernel trace + MFMA counters"
rocprofv3 --kernel-trace --stats -d $D/trace_bf -- python3 bench.py --arith bf16x3 --steps 10 --warmup 2 $LEAN > $D/bf_bench.json 2> $D/bf_trace.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $D/pmc_mfma_bf -- python3 bench.py --arith bf16x3 $ARGS > $D/pmc_mfma_bf.json 2> $D/pmc_mfma_bf.err
echo "[measure] LPIPS(vgg) loop: kernel trace + MFMA counters"
VARGS="--lpips-net vgg --batch 16 $LEAN"
rocprofv3 --kernel-trace --stats -d $D/trace_vgg -- python3 bench.py $VARGS --steps 6 --warmup 1 > $D/vgg_bench.json 2> $D/vgg_trace.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $D/pmc_mfma_vgg -- python3 bench.py $VARGS --no-graph --steps 1 --warmup 1 > $D/pmc_mfma_vgg.json 2> $D/pmc_mfma_vgg.err
echo "[measure] config 3 (Wing + FaceNet + LPIPS + MSE over a list of targets): kernel trace"
rocprofv3 --kernel-trace --stats -d $D/trace_c3 -- python3 bench.py --workload config3 --config3-targets 2 --config3-steps 64 > $D/c3_bench.json 2> $D/c3_trace.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $D/pmc_mfma_c3 -- python3 bench.py --workload config3 --config3-targets 1 --config3-steps 16 --no-graph > $D/pmc_mfma_c3.json 2> $D/pmc_mfma_c3.err
echo "[measure] reduce"
bash tools/refresh_profiles.sh $D $B $R
{ echo "== one target (n = 1), the last step of the gradient-mode leg: python tools/grad_step_trace.py <rocprofv3 --kernel-trace db>";
  python3 tools/grad_step_trace.py $(ls -t $(find $D/gtrace1 -name "*_results.db") | head -1) 60;
  echo; echo "== 16 targets in lockstep (n = 16)";
  python3 tools/grad_step_trace.py $(ls -t $(find $D/gtrace8 -name "*_results.db") | head -1) 60; } > profiles/${R}_gradient_step_trace.txt
python3 tools/grad_step_trace.py $(ls -t $(find $D/gtrace1 -name "*_results.db") | head -1) --ordered > profiles/${R}_gradient_step_ordered.txt
python3 tools/rocpd_stats.py $(ls -t $(find $D/trace_bf -name "*_results.db") | head -1) --loop-only > profiles/${R}_bf16x3_kernel_stats_loop.txt
python3 tools/iter_trace.py $(ls -t $(find $D/trace_bf -name "*_results.db") | head -1) > profiles/${R}_bf16x3_iteration_trace.txt 2>&1
python3 tools/pmc_mfma.py $D/pmc_mfma_bf --json profiles/${R}_bf16x3_pmc_mfma.json > profiles/${R}_bf16x3_pmc_mfma.txt
cp $D/bf_bench.json profiles/${R}_bf16x3_bench.json
python3 tools/rocpd_stats.py $(ls -t $(find $D/trace_vgg -name "*_results.db") | head -1) --loop-only > profiles/${R}_vgg_kernel_stats_loop.txt
python3 tools/pmc_mfma.py $D/pmc_mfma_vgg --json profiles/${R}_vgg_pmc_mfma.json > profiles/${R}_vgg_pmc_mfma.txt
cp $D/vgg_bench.json profiles/${R}_vgg_bench.json
python3 tools/rocpd_stats.py $(ls -t $(find $D/trace_c3 -name "*_results.db") | head -1) > profiles/${R}_config3_kernel_stats.txt
cp $D/c3_bench.json profiles/${R}_config3_bench.json
python3 tools/pmc_mfma.py $D/pmc_mfma_c3 --json profiles/${R}_config3_pmc_mfma.json > profiles/${R}_config3_pmc_mfma.txt
mkdir -p $D/profiles && cp profiles/${R}_* $D/profiles/
rm -rf $D/trace $D/trace1 $D/pmc_fetch $D/pmc_write $D/pmc_mfma $D/gtrace1 $D/gtrace8 $D/trace_vgg $D/pmc_mfma_vgg $D/trace_c3 $D/pmc_mfma_c3 $D/trace_bf $D/pmc_mfma_bf          # raw output: too large to travel back
echo "[measure] done"; ls -la $D/profiles
