"""How far one loop iteration is from the floors its own counters set: per kernel of the committed round's profiles,
    time per iteration (rocprofv3 kernel trace, loop only), the time its matrix pipes were busy (mfma_busy x time, MFMA counter pass) and the time
    its HBM-side traffic needs at the rate a copy reaches on this part (2 x FETCH + WRITE over HBM_TBS),
and what is left when the larger of the two is taken away:   python tools/iteration_floor.py [round tag, default r4] [HBM TB/s, default 5.0]
Reads profiles/<tag>_bench_kernel_stats_one_stream_loop.txt (or <tag>_bench_kernel_stats_loop.txt), <tag>_pmc_mfma.json, <tag>_pmc_traffic.json."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
tbs = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
rows = {}
iters = None
one = os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats_one_stream_loop.txt")          # (round 6 on: the default run is two-stream, the floors need kernels running alone)
for line in open(one if os.path.exists(one) else os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats_loop.txt")):
    m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if not m:
        continue
    name, calls, total_ms = short(m.group(1)), int(m.group(2)), float(m.group(3))
    rows[name] = [calls, total_ms]
    if name.startswith("wino3p_conv_kernel<8, false>"):
        iters = calls                                    # one launch per iteration
mf = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_mfma.json")))
tr = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")))
print(f"{'kernel':<52} {'ms/iter':>8} {'mfma ms':>8} {'hbm ms':>8} {'slack':>8}   (per iteration of 32 candidates; HBM at {tbs} TB/s)")
tot = [0.0, 0.0, 0.0, 0.0]
for name, (calls, total_ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    ms = total_ms / iters
    busy = mf.get(name, {}).get("mfma_busy", 0.0) * ms
    rec = tr.get(name)
    hbm = (rec["hbm_bytes"] * (calls / iters) / (tbs * 1e12) * 1e3) if rec else 0.0      # hbm_bytes: per launch
    floor = max(busy, hbm)
    tot = [tot[0] + ms, tot[1] + busy, tot[2] + hbm, tot[3] + ms - min(floor, ms)]
    if ms >= 0.05:
        print(f"{name[:52]:<52} {ms:8.3f} {busy:8.3f} {hbm:8.3f} {ms - min(floor, ms):8.3f}")
print(f"{'total':<52} {tot[0]:8.3f} {tot[1]:8.3f} {tot[2]:8.3f} {tot[3]:8.3f}")
print(f"matrix pipes busy {tot[1] / tot[0]:.2f} of the iteration; time above max(pipe-busy, HBM) per kernel: {tot[3] / tot[0]:.2f} of it")
