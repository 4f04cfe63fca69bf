"""Does a producer -> consumer chain run faster when its tensors fit the 256 MB memory-side cache?  Ping-pong `b = a * 1.0001` between
two (and round-robin over three) buffers of S MB each and report the effective read + write rate per size (GPU): python tools/mall_probe.py"""
import torch
def bench(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 1024):
    n = mb * (1 << 20) // 4
    for nb in (2, 3):
        bufs = [torch.randn(n, device="cuda") for _ in range(nb)]
        def chain():
            for i in range(6):
                torch.mul(bufs[i % nb], 1.0001, out=bufs[(i + 1) % nb])
        t = bench(chain, max(3, 2048 // mb)) / 6
        print(f"S = {mb:5d} MB x {nb} buffers: {2 * n * 4 / t / 1e12:5.2f} TB/s (read + write), {t * 1e6:7.1f} us per pass", flush=True)
