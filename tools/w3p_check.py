"""Persistent form-3 Winograd (wino3p_conv_kernel) against the one-shot kernel on the same inputs, then timings at the literal loop's
1024^2 shapes: python tools/w3p_check.py [--time-only]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
L = _lib.lib()
torch.manual_seed(0)
def run(shape_code, fn):
    _lib.check(L.mgf_winograd3_force_shape(shape_code)); out = fn(); torch.cuda.synchronize(); _lib.check(L.mgf_winograd3_force_shape(0)); return out.clone()
worst = 0.0
if "--time-only" not in sys.argv:
    for (n, c, co, h, w) in ((2, 32, 32, 64, 64), (1, 32, 32, 100, 96), (3, 32, 64, 32, 512), (1, 32, 32, 16, 1024), (2, 32, 32, 8, 2048)):
        x = torch.randn(n, c, h, w, device="cuda")
        wt = torch.randn(co, c, 3, 3, device="cuda") / (3 * c ** 0.5)
        s, d = torch.rand(n, c, device="cuda") + 0.5, torch.rand(n, co, device="cuda") + 0.5
        noise, bias, st = torch.randn(n, h * w, device="cuda"), torch.randn(co, device="cuda"), torch.tensor([0.3], device="cuda")
        low, resid = torch.randn(n, co, h // 2, w // 2, device="cuda"), torch.randn(n, co, h, w, device="cuda")
        u2 = cv.winograd2_weights(wt)
        eps = {"plain": _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.4),
               "resid": _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.4, residual=resid),
               "const-noise": _lib.make_epilogue(bias=bias, noise=noise[:1].contiguous(), noise_strength=st, noise_n=1, act="lrelu", alpha=0.2, gain=0.7),
               "relu-nobias": _lib.make_epilogue(act="relu")}
        cases = {k: (lambda ep=ep: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep)) for k, ep in eps.items()}
        cases["low"] = lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=eps["plain"], residual_low=low)
        cases["none"] = lambda: cv.winograd2_forward(x, u2)
        cases["scale-only"] = lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d)
        if co == 32:
            rw, rb = torch.randn(n, 3, co, device="cuda"), torch.randn(3, device="cuda")
            cases["rgb"] = lambda: cv.winograd2_rgb_forward(x, u2, rw, rb, torch.empty(n, 3, h, w, device="cuda"), in_scale=s, out_scale=d)
        for name, fn in cases.items():
            ref, got = run(11, fn), run(31, fn)
            err = float((ref - got).abs().max() / ref.abs().max())
            worst = max(worst, err)
            print(f"n={n} c={c}->{co} {h}x{w} {name:<12} rel err {err:.2e} {'OK' if err < 2e-5 else 'MISMATCH'}", flush=True)
    print("worst", worst)
    if worst >= 2e-5:
        sys.exit(1)
n, res, c = int(os.environ.get("MGF_N", "25")), 1024, 32
x = torch.randn(n, c, res, res, device="cuda")
w = torch.randn(c, c, 3, 3, device="cuda") / (3 * c ** 0.5)
s, d = torch.rand(n, c, device="cuda") + 0.5, torch.rand(n, c, device="cuda") + 0.5
noise, bias = torch.randn(n, res * res, device="cuda"), torch.randn(c, device="cuda")
st = torch.tensor([0.1], device="cuda")
low = torch.randn(n, c, res // 2, res // 2, device="cuda")
ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.4)
u2 = cv.winograd2_weights(w)
out = torch.empty_like(x)
rgb_w, rgb_b, rgb = torch.randn(n, 3, c, device="cuda"), torch.randn(3, device="cuda"), torch.empty(n, 3, res, res, device="cuda")
def timed(name, code, fn):
    _lib.check(L.mgf_winograd3_force_shape(code))
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    _lib.check(L.mgf_winograd3_force_shape(0))
    print(f"{name:<20} {'persistent' if code == 31 else 'one-shot':<10} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us", flush=True)
for code in (11, 31):
    timed("conv1 + skip", code, lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, out=out, residual_low=low))
    timed("conv1 plain ep", code, lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, out=out))
    timed("conv_last + ToRGB", code, lambda: cv.winograd2_rgb_forward(x, u2, rgb_w, rgb_b, rgb, in_scale=s, out_scale=None))
