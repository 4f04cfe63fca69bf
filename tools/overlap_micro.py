"""Does a memory-bound chain (LPIPS-like 1x1 layers + tap distances) hide behind an MFMA-bound kernel (the 1024^2 transposed conv) when the two run on
different streams?  python tools/overlap_micro.py   -> sequential vs concurrent wall time"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = 32
x = torch.randn(n, 64, 512, 512, device="cuda"); s = torch.rand(n, 64, device="cuda") + 0.5; d = torch.rand(n, 32, device="cuda") + 0.5
pc = cv.pack_weights(torch.randn(32, 64, 3, 3, device="cuda") / 24)
y = torch.randn(n, 64, 255, 255, device="cuda")
pq = cv.pack_weights(torch.randn(16, 64, 1, 1, device="cuda") / 8)
sq = torch.empty(n, 16, 255, 255, device="cuda")
def compute():
    cv.tconv3x3s2_forward(x, pc, in_scale=s, out_scale=d)
def memory():
    for _ in range(20): cv.conv_forward(y, pq, out=sq)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tc, tm = timeit(compute), timeit(memory)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa): compute()
    with torch.cuda.stream(sb): memory()
    cur.wait_stream(sa); cur.wait_stream(sb)
tb = timeit(both)
print(f"compute alone {tc:.3f} ms, memory chain alone {tm:.3f} ms, sum {tc + tm:.3f}, concurrent {tb:.3f} ms")
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    both(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st): both()
    tg = timeit(g.replay)
print(f"concurrent under graph replay {tg:.3f} ms")
