"""Calibrate achievable HBM bandwidth on this device with simple streaming ops (GPU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from morphganformer_amd.torch_utils.ops import bias_act, upfirdn2d


def bench(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for mb in (134, 536, 2144):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    t = bench(lambda: y.copy_(x)); print(f"torch copy {mb} MB: {2 * n * 4 / t / 1e12:.2f} TB/s")
    t = bench(lambda: torch.add(x, 1.0, out=y)); print(f"torch add  {mb} MB: {2 * n * 4 / t / 1e12:.2f} TB/s")
x = torch.randn(1, 32, 1024, 1024, device="cuda"); b = torch.randn(32, device="cuda")
t = bench(lambda: bias_act.bias_act(x, b, act="lrelu")); print(f"mgf bias_act [1,32,1024,1024]: {2 * x.numel() * 4 / t / 1e12:.2f} TB/s")
f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
t = bench(lambda: upfirdn2d.upfirdn2d(x, f, padding=[2, 1, 2, 1], gain=1.0)); print(f"mgf upfirdn2d 4x4 [1,32,1024,1024]: {2 * x.numel() * 4 / t / 1e12:.2f} TB/s")
