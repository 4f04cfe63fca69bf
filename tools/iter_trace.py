"""Print the kernels of the last full projection iteration in a rocprofv3 rocpd database + per-kernel totals.
    python tools/iter_trace.py gpurun_out/prof/x_results.db [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = db.execute("select name, end-start, grid_x, grid_y, grid_z, lds_size, vgpr_count, workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "perturb" in r[0]]
a, b = idx[-2], idx[-1]
tot, agg = 0, {}
for r in rows[a:b]:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    short = nm.split("(")[0]
    if r[1] > thr * 1e3:
        print(f"{short:<46} {r[1] / 1e3:8.1f} us grid=({r[2] // max(r[7], 1)},{r[3]},{r[4]}) lds={r[5]} vgpr={r[6]}")
    tot += r[1]
    v = agg.setdefault(short, [0, 0])
    v[0] += r[1]; v[1] += 1
print(f"iteration total {tot / 1e6:.3f} ms over {b - a} kernels")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:<50} {v[0] / 1e3:8.1f} us  x{v[1]}")
