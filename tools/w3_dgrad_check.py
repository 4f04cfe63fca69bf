"""Winograd vs tap-list kernel for a data-gradient-shaped 3x3 conv with an in-place residual (GPU check): python tools/w3_dgrad_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
for n, cin, cout, h, w in ((8, 128, 32, 127, 127), (8, 256, 64, 63, 63), (2, 128, 32, 17, 17), (2, 256, 64, 8, 8), (2, 128, 32, 16, 16), (2, 64, 32, 34, 34), (2, 64, 32, 33, 35)):
    torch.manual_seed(0)
    x = torch.randn(n, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    r = torch.randn(n, cout, h, w, device="cuda")
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1) + r.double()
    pc = cv.pack_weights(wt)
    a = r.clone(); cv.conv_forward(x, pc, pad=(1, 1), epilogue=_lib.make_epilogue(residual=a), out=a)
    u = cv.winograd2_weights(wt, 1.0)
    b = r.clone(); cv.winograd_forward(x, u, epilogue=_lib.make_epilogue(residual=b), out=b)
    c = torch.empty_like(r); cv.winograd_forward(x, u, epilogue=_lib.make_epilogue(residual=r), out=c)
    e = lambda t: float((t.double() - ref).abs().max() / ref.abs().max())
    print(f"n {n} {cin}->{cout} {h}x{w}: taps {e(a):.2e}  winograd in place {e(b):.2e}  winograd out of place {e(c):.2e}", flush=True)
