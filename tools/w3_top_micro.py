"""The two 1024^2 form-3 Winograd launches of the literal loop at 25 samples -- conv1 (noise / bias / lrelu epilogue + the skip branch
up-sampled in the epilogue) and conv_last + ToRGB -- for ablation builds (tools/build_exp.sh x "-DMGF_W3X=..." wino3.hip): python tools/w3_top_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n, res, c = int(os.environ.get("MGF_N", "32")), 1024, 32
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
def timed(name, fn):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('MGF_LIB_PATH', 'default'):<34} {name:<18} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us", flush=True)
timed("conv1 + skip", lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, out=out, residual_low=low))
timed("conv1 plain ep", lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, out=out))
timed("conv_last + ToRGB", lambda: cv.winograd2_rgb_forward(x, u2, rgb_w, rgb_b, rgb, in_scale=s, out_scale=None))
