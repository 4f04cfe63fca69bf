"""MFMA-pipe utilisation and shader clock per kernel from one rocprofv3 counter pass:

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d OUT -- python3 bench.py ... --no-graph
    python tools/pmc_mfma.py OUT [--json profiles/rN_pmc_mfma.json]        (the JSON is what bench.py reports as roofline.mfma_busy)

Only the projection iterations are counted (first perturb_kernel .. last select_kernel).  Normalisation on MI355X: rocprofv3 sums
GRBM_GUI_ACTIVE over the 8 XCDs and SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs, so
    clock = GRBM_GUI_ACTIVE / 8 / duration            MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
(a kernel that kept every MFMA pipe busy every cycle would read 1.0; achieved TFLOP/s = utilisation x 157.3 x clock / 2.4 GHz).
"""
import csv
import glob
import os
import sys
from collections import defaultdict

XCDS, SIMDS = 8, 1024


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
    disp = defaultdict(dict)
    for f in files:
        for r in csv.DictReader(open(f, newline="")):
            d = disp[(f, int(r["Dispatch_Id"]))]
            d["name"] = r["Kernel_Name"]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = defaultdict(lambda: defaultdict(float))
    keys = sorted(disp)
    first = min((k for k in keys if "perturb_kernel" in disp[k]["name"]), default=None)
    last = max((k for k in keys if "select_kernel" in disp[k]["name"]), default=None)
    for key in keys:
        if first is None or last is None or key < first or key > last:
            continue
        d = disp[key]
        name = d["name"]
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        a = agg[short]
        for c in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "ns"):
            a[c] += d.get(c, 0.0)
        a["launches"] += 1
    print(f"{'kernel':<44}{'launches':>9}{'total_ms':>10}{'clock_GHz':>11}{'mfma_util':>11}")
    table = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        if a["ns"] <= 0:
            continue
        cyc = a["GRBM_GUI_ACTIVE"] / XCDS
        util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(cyc * SIMDS, 1)
        print(f"{k[:44]:<44}{int(a['launches']):>9}{a['ns'] / 1e6:>10.2f}{cyc / a['ns']:>11.2f}{util:>11.3f}")
        table[k.strip()] = {"launches": int(a["launches"]), "total_ms": round(a["ns"] / 1e6, 3), "clock_ghz": round(cyc / a["ns"], 3),
                            "mfma_busy": round(util, 4)}
    if "--json" in sys.argv:
        import json
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(table, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
