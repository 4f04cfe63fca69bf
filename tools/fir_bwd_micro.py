"""The blur gradient (pad 2, output one larger) on the separable kernel vs the generic tiled one (GPU): python tools/fir_bwd_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import conv as cv
from morphganformer_amd.torch_utils.ops import upfirdn2d
f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
n = 8
for c, r in ((32, 1024), (64, 512), (128, 256), (256, 128)):
    x = torch.randn(n, c, r, r, device="cuda"); out = torch.empty(n, c, r + 1, r + 1, device="cuda")
    for sep in (True, False):
        fn = lambda: cv.upfirdn_into(out, x, f, up=1, pad=(2, 2, 2, 2), gain=4.0, flip=True, separable=sep)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"c {c} r {r} sep {sep}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us", flush=True)
