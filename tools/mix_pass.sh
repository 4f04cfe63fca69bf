#!/bin/bash
# SQ instruction-mix / wait counters over one bench iteration (two --pmc passes, reduced by tools/pmc_mix.py):
#   gpurun -- 'bash tools/mix_pass.sh gpurun_out/mix r5 ["--arith bf16x3"]'   ->  gpurun_out/mix/r5_pmc_mix.txt
set -e
D=${1:-gpurun_out/mix}; R=${2:-r5}; EXTRA=${3:-}; ROOT=$(pwd); mkdir -p $D        # EXTRA: further bench.py flags, e.g. "--arith bf16x3"
ARGS="--steps 1 --warmup 1 --batch 32 --no-graph --no-cpu-baseline --gradient-steps 0 --targets 0 --landmark-callback none --objectives 0 --config4 0 --config5-targets 0 --bf16x3-leg 0 $EXTRA"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $D/p1 -- python3 bench.py $ARGS > $D/p1.json 2> $D/p1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $D/p2 -- python3 bench.py $ARGS > $D/p2.json 2> $D/p2.err
python3 tools/pmc_mix.py $D/p1 $D/p2 > $D/${R}_pmc_mix.txt
rm -rf $D/p1 $D/p2
head -12 $D/${R}_pmc_mix.txt | cut -c1-400
