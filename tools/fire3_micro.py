"""The eight Fire expand3x3 layers of SqueezeNet1.1 at 1024^2 on the form-3 Winograd kernel, workgroup shape pinned: python tools/fire3_micro.py [n]"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = _lib.lib()
tot = {}
for cin, ex, side in ((16, 64, 255), (16, 64, 255), (32, 128, 127), (32, 128, 127), (48, 192, 63), (48, 192, 63), (64, 256, 63), (64, 256, 63)):
    x = torch.randn(n, cin, side, side, device="cuda")
    wt = torch.randn(ex, cin, 3, 3, device="cuda") / math.sqrt(9 * cin)
    b = torch.randn(ex, device="cuda")
    u2 = cv.winograd2_weights(wt)
    y = torch.empty(n, 2 * ex, side, side, device="cuda")
    line = f"{cin:3d}->{ex:3d} at {side:3d}^2:"
    for shape in (0, 11, 21, 12):
        _lib.check(L.mgf_winograd3_force_shape(shape))
        try:
            fn = lambda: cv.winograd2_forward(x, u2, epilogue=_lib.make_epilogue(bias=b, act="relu"), out=y, out_choff=ex)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            tot[shape] = tot.get(shape, 0) + us
            line += f"  shape {shape:2d}: {us:7.1f} us"
        except Exception as e:
            line += f"  shape {shape:2d}: {type(e).__name__}"
    print(line, flush=True)
_lib.check(L.mgf_winograd3_force_shape(0))
print("totals:", {k: round(v, 1) for k, v in tot.items()})
