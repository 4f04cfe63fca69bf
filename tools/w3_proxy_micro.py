"""Proxy for a Winograd transposed conv: form-3 F(2x2,3x3) on the transposed convs' INPUT shapes.  A F(2,2) transposed conv needs 25 products per
2x2 input block where form 3 needs 16 per 2x2 output block, so 25/16 of these times estimate it: python tools/w3_proxy_micro.py [n]"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for cin, cout, res in ((64, 32, 512), (128, 64, 256), (256, 128, 128), (512, 256, 64), (512, 512, 32)):
    x = torch.randn(n, cin, res, res, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") / math.sqrt(9 * cin)
    u2 = cv.winograd2_weights(wt)
    s, d = torch.rand(n, cin, device="cuda") + 0.5, torch.rand(n, cout, device="cuda") + 0.5
    out = torch.empty(n, cout, res, res, device="cuda")
    fn = lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, out=out)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"{cin:3d}->{cout:3d} at {res:3d}^2 n {n}: form 3 {us:7.1f} us   x 25/16 = {us * 25 / 16:7.1f} us", flush=True)
