"""Kernels of the last gradient-mode step in a rocprofv3 rocpd database (between the last two launches of the mapping network's BACKWARD
kernel: one per step, also with several targets in lockstep), per-kernel totals:
    python tools/grad_step_trace.py gpurun_out/x/trace/*/NNN_results.db [rows]
    python tools/grad_step_trace.py DB --ordered        # every launch of the step in order: start offset, duration, gap to the previous end"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, end-start, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "mapping_backward" in r[0]]
a, b = idx[-2], idx[-1]


def short(nm):
    return nm.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]


if "--ordered" in sys.argv:
    t0, prev_end = rows[a][2], rows[a][2]
    print(f"step: {b - a} kernels, wall {(rows[b][2] - rows[a][2]) / 1e6:.3f} ms")
    for i, r in enumerate(rows[a:b]):
        print(f"{i:4d} {(r[2] - t0) / 1e3:9.1f} us  dur {r[1] / 1e3:7.1f}  gap {(r[2] - prev_end) / 1e3:6.1f}  {short(r[0])}")
        prev_end = r[3]
    sys.exit(0)
tot, agg = 0, {}
for r in rows[a:b]:
    nm = short(r[0])
    v = agg.setdefault(nm, [0, 0]); v[0] += r[1]; v[1] += 1; tot += r[1]
print(f"step: {b - a} kernels, kernel time {tot / 1e6:.3f} ms, wall {(rows[b][2] - rows[a][2]) / 1e6:.3f} ms")
nrows = [x for x in sys.argv[2:] if x.isdigit()]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(nrows[0]) if nrows else 45]:
    print(f"{k:<66} {v[0] / 1e3:8.1f} us x{v[1]:3d}  avg {v[0] / 1e3 / v[1]:7.1f}")
