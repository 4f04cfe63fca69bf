"""The skip branch's gradient (down 2, pad 1) at the sizes of a lockstep-8 gradient step (GPU): python tools/fir_down_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import conv as cv
from morphganformer_amd.torch_utils.ops import upfirdn2d
f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
n = 8
for c, r in ((32, 1024), (64, 512), (128, 256), (256, 128)):
    x = torch.randn(n, c, r, r, device="cuda"); out = torch.empty(n, c, r // 2, r // 2, device="cuda")
    fn = lambda: cv.upfirdn_into(out, x, f, up=1, down=2, pad=(1, 1, 1, 1), gain=4.0, flip=True)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"c {c} r {r}: {us:7.1f} us  {(x.numel() + out.numel()) * 4 / us / 1e6:5.2f} TB/s", flush=True)
