"""The two one-shot form-3 launches with the fused half-resolution skip (conv1 of the 256^2 / 512^2 blocks) at MGF_N samples: python tools/w3_low_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = int(os.environ.get("MGF_N", "32"))
for res, c in ((256, 128), (512, 64)):
    x = torch.randn(n, c, res, res, device="cuda")
    w = torch.randn(c, c, 3, 3, device="cuda") / (3 * c ** 0.5)
    s, d = torch.rand(n, c, device="cuda") + 0.5, torch.rand(n, c, device="cuda") + 0.5
    noise, bias = torch.randn(n, res * res, device="cuda"), torch.randn(c, device="cuda")
    st = torch.tensor([0.1], device="cuda")
    low = torch.randn(n, c, res // 2, res // 2, device="cuda")
    ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.4)
    u2 = cv.winograd2_weights(w)
    out = torch.empty_like(x)
    fn = lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, out=out, residual_low=low)
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('MGF_LIB_PATH', 'default')[-24:]:<24} res {res} c {c} n {n}: conv1 + skip {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us", flush=True)
