#!/bin/bash
# L2-miss fetch (2 x FETCH_SIZE) and WRITE_SIZE per launch of the transposed-conv kernel on the generator's five big up-sampling layers at 32 samples:
#   bash tools/tconv_fetch_layers.sh OUT
set -e
D=${1:-gpurun_out/tconv_fetch}; R=$(pwd); mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  echo "[tconv_fetch] $c pass"
  MGF_MICRO_N=32 timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $R/$D/$c -- python3 $R/tools/conv_micro.py r64_tconv r128_tconv r256_tconv r512_tconv r1024_tconv > $R/$D/$c.log 2>&1
done
cd $R
python3 - $D <<'PY'
import csv, glob, sys
shapes = [("r64_tconv", 512, 512, 32), ("r128_tconv", 512, 256, 64), ("r256_tconv", 256, 128, 128), ("r512_tconv", 128, 64, 256), ("r1024_tconv", 64, 32, 512)]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"{sys.argv[1]}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if "conv_taps_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-45:], float(r["Counter_Value"])))
    rows.sort()
    per = len(rows) // len(shapes)
    vals[c] = [rows[i * per:(i + 1) * per] for i in range(len(shapes))]
for i, (name, ci, co, res) in enumerate(shapes):
    f = vals["FETCH_SIZE"][i]; w = vals["WRITE_SIZE"][i]
    fetch = 2 * sum(g[2] for g in f) / len(f) * 1024 / 1e9
    write = sum(g[2] for g in w) / len(w) * 1024 / 1e9
    inp = 32 * ci * res * res * 4 / 1e9
    out = 32 * co * (2 * res + 1) * (2 * res + 1) * 4 / 1e9
    print(f"{name:12s} {ci:3d}->{co:3d} in {res:3d}^2: {f[0][1]}  fetch {fetch:6.2f} GB (input {inp:5.2f} GB: {fetch / inp:.2f} x)   write {write:6.2f} GB (output {out:5.2f} GB: {write / out:.2f} x)")
PY
rm -rf $D/FETCH_SIZE $D/WRITE_SIZE
