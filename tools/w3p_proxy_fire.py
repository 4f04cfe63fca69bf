import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from morphganformer_amd import _lib, conv as cv
L = _lib.lib()
n = 32
for cin, cout, res in ((32, 128, 128), (32, 64, 256), (32, 32, 256)):
    x = torch.randn(n, cin, res, res, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") / math.sqrt(9 * cin)
    u = cv.winograd_pack(wt, 1.0, res)
    bias = torch.randn(cout, device="cuda")
    ep = _lib.make_epilogue(bias=bias, act="relu")
    out = torch.empty(n, cout, res, res, device="cuda")
    for shape in (11, 31, 11, 31):
        _lib.check(L.mgf_winograd3_force_shape(shape))
        fn = lambda: cv.winograd2_forward(x, u, epilogue=ep, out=out)
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"n {n} {cin:3d}->{cout:3d} at {res:3d}^2 shape {shape}: {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us", flush=True)
    _lib.check(L.mgf_winograd3_force_shape(0))
