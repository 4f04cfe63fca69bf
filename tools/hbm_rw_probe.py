"""Write-only / copy / read-only HBM rates with torch streaming ops (GPU): python tools/hbm_rw_probe.py"""
import torch
def bench(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for gb in (0.84, 3.36):
    n = int(gb * 1e9 / 4)
    y = torch.empty(n, device="cuda"); x = torch.randn(n, device="cuda")
    t = bench(lambda: y.fill_(1.0)); print(f"fill  {gb} GB: {n*4/t/1e12:.2f} TB/s ({t*1e6:.0f} us)")
    t = bench(lambda: y.copy_(x)); print(f"copy  {gb} GB: {2*n*4/t/1e12:.2f} TB/s total ({t*1e6:.0f} us)")
    t = bench(lambda: x.sum()); print(f"read  {gb} GB: {n*4/t/1e12:.2f} TB/s ({t*1e6:.0f} us)")
