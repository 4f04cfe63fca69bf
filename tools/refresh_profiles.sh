#!/bin/bash
# Reduce one measurement pass (gpurun_out/<dir> with bench.json, trace/, pmc_fetch/, pmc_write/, pmc_mfma/) into profiles/<round>_*.
# usage: tools/refresh_profiles.sh gpurun_out/r2m [steps_per_forward] [round]
set -e
D=$1; B=${2:-32}; R=${3:-r5}
cd "$(dirname "$0")/.."
DB=$(ls -t $(find $D/trace -name "*_results.db") | head -1)      # the newest database: a re-used directory may hold an older one
python tools/rocpd_stats.py $DB > profiles/${R}_bench_kernel_stats.txt
python tools/rocpd_stats.py $DB --loop-only > profiles/${R}_bench_kernel_stats_loop.txt
DB1=$DB; if [ -d $D/trace1 ]; then DB1=$(ls -t $(find $D/trace1 -name "*_results.db") | head -1); fi      # one-stream trace (measure_pass.sh) for the per-iteration analysis
python tools/rocpd_stats.py $DB1 --loop-only > profiles/${R}_bench_kernel_stats_one_stream_loop.txt
python tools/iter_trace.py $DB1 > profiles/${R}_iteration_trace.txt 2>&1
python tools/pmc_mfma.py $D/pmc_mfma --json profiles/${R}_pmc_mfma.json > profiles/${R}_pmc_mfma.txt
cp $D/bench.json profiles/${R}_bench.json
CAL=$(sed -n '/^calibration/,$p' profiles/r1_pmc_traffic.txt)
python tools/pmc_traffic.py $D/pmc_fetch $D/pmc_write --json profiles/${R}_pmc_traffic.json --steps-per-forward $B --loop-only > profiles/${R}_pmc_traffic.txt
printf "\n%s\n" "$CAL" >> profiles/${R}_pmc_traffic.txt
