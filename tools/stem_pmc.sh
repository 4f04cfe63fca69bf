#!/bin/bash
# SQ wait / instruction-mix counters over the LPIPS stem alone: bash tools/stem_pmc.sh OUTDIR [N]   (two --pmc passes, then tools/pmc_mix.py)
set -e
D=${1:-gpurun_out/stem_pmc}; N=${2:-32}; R=$(pwd); mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $R/$D/p1 -- python3 $R/tools/stem_micro.py $N > $R/$D/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $R/$D/p2 -- python3 $R/tools/stem_micro.py $N > $R/$D/p2.log 2>&1
cd $R
python3 tools/pmc_mix.py $D/p1 $D/p2 --match stem > $D/stem_mix.txt
rm -rf $D/p1 $D/p2
cat $D/stem_mix.txt
