"""Per-kernel SQ instruction mix / wait breakdown from rocprofv3 counter CSVs (any set of counters, one or more passes):
    python tools/pmc_mix.py DIR [DIR ...] [--match SUBSTR]
prints, per kernel, the per-launch mean of every counter found and a few ratios (VALU instructions per MFMA, wait fractions)."""
import csv, glob, os, sys
from collections import defaultdict
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
agg = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if match and match not in name:
                continue
            a = agg[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1].add((f, r["Dispatch_Id"]))
for name, cs in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_INSTS_VALU", [0]))[0]):
    m = {k: v[0] / max(len(v[1]), 1) for k, v in cs.items()}
    line = f"{name[:60]:<60} launches {max(len(v[1]) for v in cs.values()):4d}"
    for k in sorted(m):
        line += f"  {k.replace('SQ_', '')}={m[k]:.3g}"
    if "SQ_INSTS_MFMA" in m and m["SQ_INSTS_MFMA"] > 0:
        line += f"  | VALU/MFMA={m.get('SQ_INSTS_VALU', 0) / m['SQ_INSTS_MFMA']:.2f} LDS/MFMA={m.get('SQ_INSTS_LDS', 0) / m['SQ_INSTS_MFMA']:.2f} SALU/MFMA={m.get('SQ_INSTS_SALU', 0) / m['SQ_INSTS_MFMA']:.2f} VMEM/MFMA={m.get('SQ_INSTS_VMEM', 0) / m['SQ_INSTS_MFMA']:.2f}"
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
        w = m["SQ_WAVE_CYCLES"]
        line += f"  | wait_any={m.get('SQ_WAIT_ANY', 0) / w:.2f} wait_inst={m.get('SQ_WAIT_INST_ANY', 0) / w:.2f} active={m.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f} active_valu={m.get('SQ_ACTIVE_INST_VALU', 0) / w:.2f} active_lds={m.get('SQ_ACTIVE_INST_LDS', 0) / w:.2f}"
    print(line)
