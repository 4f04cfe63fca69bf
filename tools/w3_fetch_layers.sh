#!/bin/bash
# L2-miss fetch (2 x FETCH_SIZE) per launch of the one-shot form-3 kernel on the proxy micro's five shapes: bash tools/w3_fetch_layers.sh OUT
set -e
D=${1:-gpurun_out/w3_fetch_layers}; R=$(pwd); mkdir -p $D
cd /tmp && export TMPDIR=/tmp
echo "[w3_fetch_layers] FETCH_SIZE pass"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$D/f -- python3 $R/tools/w3_proxy_micro.py 32 > $R/$D/f.log 2>&1
cd $R
python3 - $D <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(f"{sys.argv[1]}/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if "wino3" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-45:], float(r["Counter_Value"])))
rows.sort()
shapes = [(64, 32, 512), (128, 64, 256), (256, 128, 128), (512, 256, 64), (512, 512, 32)]
per = len(rows) // len(shapes)
for i, (ci, co, res) in enumerate(shapes):
    grp = rows[i * per:(i + 1) * per]
    fetch = 2 * sum(g[2] for g in grp) / len(grp) * 1024 / 1e9
    inp = 32 * ci * res * res * 4 / 1e9
    print(f"{ci:3d}->{co:3d} at {res:3d}^2: {grp[0][1]}  fetch {fetch:6.2f} GB per launch, input {inp:5.2f} GB  ({fetch / inp:.2f} x)")
PY
rm -rf $D/f
