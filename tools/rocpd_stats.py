"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (markdown/CSV-ish text).

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--per-launch KERNEL_SUBSTR] [--loop-only]

--loop-only keeps the dispatches from the first perturb_kernel through the last select_kernel (start order), i.e. the projection
iterations: that is the population bench.py's in-process event timing averages over (no batch-1 set-up, no generator-only leg).
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    if "--loop-only" in sys.argv:
        agg = {}
        allk = cur.execute(f"select {name_col}, end-start, start from kernels order by start").fetchall()
        t0 = min((st for nm, _, st in allk if "perturb_kernel" in nm), default=None)
        t1 = max((st for nm, _, st in allk if "select_kernel" in nm), default=None)
        for name, dur, st in allk:
            if t0 is not None and t1 is not None and t0 <= st <= t1:
                a = agg.setdefault(name, [0, 0, 1 << 62, 0])
                a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
        rows = sorted(((k, v[0], v[1], v[1] / v[0], v[2], v[3]) for k, v in agg.items()), key=lambda r: -r[2])
    else:
        rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':<100} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{name[:100]:<100} {n:>6} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} {100 * tot / total:>6.2f}")
    print(f"TOTAL kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    if "--per-launch" in sys.argv:
        sub = sys.argv[sys.argv.index("--per-launch") + 1]
        extra = [c for c in ("grid_size", "workgroup_size", "lds_size", "vgpr_count", "sgpr_count", "grid_x", "grid_y", "grid_z") if c in cols]
        q = f"select {name_col}, end-start, {', '.join(extra)} from kernels where {name_col} like ? order by start"
        for r in cur.execute(q, (f"%{sub}%",)).fetchall()[-80:]:
            print(r[0][:60], f"{r[1] / 1e3:.1f}us", dict(zip(extra, r[2:])))


if __name__ == "__main__":
    main()
