#!/bin/bash
# Reduce one measurement pass (gpurun_out/<dir> with bench.json, trace/, pmc_fetch/, pmc_write/, pmc_mfma/) into profiles/r1_*.
# usage: tools/refresh_profiles.sh gpurun_out/r1m [steps_per_forward]
set -e
D=$1; B=${2:-25}
cd "$(dirname "$0")/.."
DB=$(find $D/trace -name "*_results.db" | head -1)
python tools/rocpd_stats.py $DB > profiles/r1_bench_kernel_stats.txt
python tools/rocpd_stats.py $DB --loop-only > profiles/r1_bench_kernel_stats_loop.txt
python tools/iter_trace.py $DB > profiles/r1_iteration_trace.txt 2>&1
python tools/pmc_mfma.py $D/pmc_mfma > profiles/r1_pmc_mfma.txt
cp $D/bench.json profiles/r1_bench.json
CAL=$(sed -n '/^calibration/,$p' profiles/r1_pmc_traffic.txt)
python tools/pmc_traffic.py $D/pmc_fetch $D/pmc_write --json profiles/r1_pmc_traffic.json --steps-per-forward $B --loop-only > profiles/r1_pmc_traffic.txt
printf "\n%s\n" "$CAL" >> profiles/r1_pmc_traffic.txt
