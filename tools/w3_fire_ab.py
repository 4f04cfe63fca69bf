"""One-shot form-3 Winograd shapes on the LPIPS(squeeze) Fire expand3x3 layers (K = 16 .. 64 channels, odd maps) at 32 samples: python tools/w3_fire_ab.py"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for cin, cout, res in ((16, 64, 255), (32, 128, 127), (48, 192, 63), (64, 256, 63)):
    x = torch.randn(n, cin, res, res, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") / math.sqrt(9 * cin)
    u = cv.winograd_pack(wt, 1.0, res)
    bias = torch.randn(cout, device="cuda")
    ep = _lib.make_epilogue(bias=bias, act="relu")
    out = torch.empty(n, 2 * cout, res, res, device="cuda")          # the concat buffer: the 3x3 branch writes channels [cout, 2 cout)
    for shape in ([int(v) for v in os.environ["MGF_FIRE_SHAPES"].split(",")] if os.environ.get("MGF_FIRE_SHAPES") else (11, 21, 12, 11, 21, 12)):
        _lib.check(L.mgf_winograd3_force_shape(shape))
        fn = lambda: cv.winograd2_forward(x, u, epilogue=ep, out=out, out_choff=cout)
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"n {n} {cin:3d}->{cout:3d} at {res:3d}^2 shape {shape}: {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us", flush=True)
    _lib.check(L.mgf_winograd3_force_shape(0))
