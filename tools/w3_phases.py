"""Form-3 Winograd on the generator's conv1 shapes with each workgroup shape pinned: python tools/w3_phases.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = int(os.environ.get("MGF_N", "25"))
for res, c in ((64, 512), (128, 256), (256, 128), (512, 64), (1024, 32)):
    x = torch.randn(n, c, res, res, device="cuda")
    w = torch.randn(c, c, 3, 3, device="cuda") / (3 * c ** 0.5)
    s, d = torch.rand(n, c, device="cuda") + 0.5, torch.rand(n, c, device="cuda") + 0.5
    noise, bias = torch.randn(n, res * res, device="cuda"), torch.randn(c, device="cuda")
    st = torch.tensor([0.1], device="cuda")
    resid = torch.randn(n, c, res, res, device="cuda")
    ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.4, residual=resid)
    u2 = cv.winograd2_weights(w)
    out = torch.empty_like(x)
    line = f"res {res:4d} c {c:3d}:"
    for shape in (21, 12, 11):
        if shape == 21 and c % 64:
            continue
        _lib.check(_lib.lib().mgf_winograd3_force_shape(shape))
        fn = lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, out=out)
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        line += f"  shape {shape}: {e0.elapsed_time(e1)/5*1e3:8.1f} us"
    print(line, flush=True)
_lib.lib().mgf_winograd3_force_shape(0)
# fused ToRGB (conv_last of the 1024^2 block)
x = torch.randn(n, 32, 1024, 1024, device="cuda")
w = torch.randn(32, 32, 3, 3, device="cuda") / (3 * 32 ** 0.5)
s, d = torch.rand(n, 32, device="cuda") + 0.5, torch.rand(n, 32, device="cuda") + 0.5
rw, rb = torch.randn(n, 3, 32, device="cuda"), torch.randn(3, device="cuda")
u2 = cv.winograd2_weights(w)
img = torch.empty(n, 3, 1024, 1024, device="cuda")
line = "conv_last+ToRGB 1024:"
for shape, form in ((0, 2), (12, 3), (11, 3)):
    cv.WINOGRAD_FORM = form
    _lib.check(_lib.lib().mgf_winograd3_force_shape(shape))
    fn = lambda: cv.winograd2_rgb_forward(x, u2, rw, rb, img, in_scale=s, out_scale=d)
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    line += f"  form {form} shape {shape}: {e0.elapsed_time(e1)/5*1e3:8.1f} us"
print(line, flush=True)
_lib.lib().mgf_winograd3_force_shape(0)
