"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (markdown/CSV-ish text).

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--per-launch KERNEL_SUBSTR]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
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
