#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the persistent Winograd kernel at 32 x 1024^2, vertical vs horizontal strips: bash tools/w3p_fetch.sh OUT
set -e
D=${1:-gpurun_out/w3p_fetch}; R=$(pwd); mkdir -p $D
cd /tmp && export TMPDIR=/tmp
# (one counter per pass: FETCH_SIZE and WRITE_SIZE in one pass abort rocprofv3 on this image -- the guide's HBM section says separate passes)
for v in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "[w3p_fetch] vert=$v $c"
    MGF_W3_VERT=$v MGF_N=32 timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $R/$D/v$v/$c -- python3 $R/tools/w3p_check.py --time-only > $R/$D/v${v}_$c.log 2>&1
  done
done
cd $R
python3 - $D <<'PY'
import csv, glob, sys, collections
for v in (1, 0):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{sys.argv[1]}/v{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if "wino3p" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(f"vert={v} {k}: launches {len(c['FETCH_SIZE'])}  fetch {2 * sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) * 1024 / 1e9:.2f} GB (2 x FETCH_SIZE)  write {sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE']) * 1024 / 1e9:.2f} GB per launch")
PY
rm -rf $D/v1 $D/v0
