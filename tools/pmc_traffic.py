"""Per-kernel HBM traffic from rocprofv3 counter-collection CSVs (one pass per counter: FETCH_SIZE costs 3 of the 4 TCC
slots and WRITE_SIZE 2, MI355X_MICROARCH.md 'rocprofv3 PMC slots').

    python tools/pmc_traffic.py FETCH_DIR WRITE_DIR [--calib CALIB_FETCH_DIR CALIB_WRITE_DIR] [--match SUBSTR] [--json OUT] [--steps-per-forward B] [--loop-only]

Prints, per kernel name, launches and the mean counter value per launch (raw KiB -> bytes), and when --calib is given
the factors (true bytes / reported bytes) measured on tools/pmc_calib.py's 1 GiB copies.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


LOOP_ONLY = False


def load(d):
    """{kernel: [values per dispatch]} for the single counter collected in directory tree d.  With --loop-only, only the
    dispatches from the first perturb_kernel through the last select_kernel are kept (the projection iterations): one-off set-up work
    at batch 1 and bench.py's generator-only leg would otherwise skew the per-launch means."""
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    per = defaultdict(lambda: defaultdict(float))
    for f in files:
        with open(f, newline="") as fh:
            rows = sorted(csv.DictReader(fh), key=lambda r: int(r.get("Dispatch_Id") or r.get("Correlation_Id")))
        ids = [int(r.get("Dispatch_Id") or r.get("Correlation_Id")) for r in rows]
        first = min((i for i, r in zip(ids, rows) if "perturb_kernel" in r["Kernel_Name"]), default=None)
        last = max((i for i, r in zip(ids, rows) if "select_kernel" in r["Kernel_Name"]), default=None)
        for i, row in zip(ids, rows):
            name = row["Kernel_Name"]
            if LOOP_ONLY and (first is None or last is None or i < first or i > last):
                continue
            key = (f, row.get("Dispatch_Id") or row.get("Correlation_Id"))
            per[name][key] += float(row["Counter_Value"])
    return {k: list(v.values()) for k, v in per.items()}


def mean(v):
    return sum(v) / max(len(v), 1)


def main():
    args = [a for a in sys.argv[1:]]
    match = None
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1]
        del args[i:i + 2]
    json_out = None
    if "--json" in args:
        i = args.index("--json")
        json_out = args[i + 1]
        del args[i:i + 2]
    global LOOP_ONLY
    if "--loop-only" in args:
        LOOP_ONLY = True
        args.remove("--loop-only")
    meta = None
    if "--steps-per-forward" in args:
        i = args.index("--steps-per-forward")
        meta = int(args[i + 1])
        del args[i:i + 2]
    calib = None
    if "--calib" in args:
        i = args.index("--calib")
        calib = (args[i + 1], args[i + 2])
        del args[i:i + 3]
    fetch, write = load(args[0]), load(args[1])
    KIB = 1024.0
    if calib:
        cf, cw = load(calib[0]), load(calib[1])
        true = float(1 << 30)
        for name in cf:
            if "elementwise" in name or "copy" in name.lower():
                r, w = mean(cf[name]) * KIB, mean(cw.get(name, [0])) * KIB
                if r > 1e8:
                    print(f"calib {name[:90]}: FETCH {r / 1e6:.1f} MB (true {true / 1e6:.1f}; factor {true / r:.3f}), "
                          f"WRITE {w / 1e6:.1f} MB (factor {true / max(w, 1):.3f})")
    print(f"{'kernel':<90} {'launches':>8} {'FETCH_MB':>10} {'WRITE_MB':>10}")
    rows = []
    for name in sorted(set(fetch) | set(write)):
        if match and match not in name:
            continue
        rows.append((name, len(fetch.get(name, [])), mean(fetch.get(name, [0])) * KIB / 1e6, mean(write.get(name, [0])) * KIB / 1e6))
    for name, n, r, w in sorted(rows, key=lambda t: -(t[2] + t[3]) * t[1]):
        print(f"{name[:90]:<90} {n:>8} {r:>10.2f} {w:>10.2f}")
    if json_out:
        import json
        import re
        table = {}
        for name, n, r, w in rows:
            short = re.sub(r"^void ", "", name)
            short = re.sub(r"\(anonymous namespace\)::", "", short)
            short = re.sub(r"\(.*$", "", short).strip()
            # FETCH_SIZE tallies 128-B requests at 64 B on gfx950 (guide + tools/pmc_calib.py: 536.9 MB reported for a 1073.7 MB
            # read, 559 MB for the dword-wide variant) -> doubled; WRITE_SIZE is exact (1073.7 / 1101 MB reported for 1073.7 MB)
            table[short] = {"launches": n, "fetch_bytes_reported": round(r * 1e6), "fetch_bytes": round(2 * r * 1e6),
                            "write_bytes": round(w * 1e6), "hbm_bytes": round((2 * r + w) * 1e6)}
        if meta is not None:
            table["_meta"] = {"steps_per_forward": meta}
        with open(json_out, "w") as fh:
            json.dump(table, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
