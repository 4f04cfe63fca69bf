"""Form-3 Winograd conv1 layers at fewer resident workgroups (MGF_W3_LDS_PAD bytes of extra LDS): python tools/w3_occupancy.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = 25
for res, c in ((64, 512), (256, 128), (512, 64), (1024, 32)):
    x = torch.randn(n, c, res, res, device="cuda")
    w = torch.randn(c, c, 3, 3, device="cuda") / (3 * c ** 0.5)
    s, d = torch.rand(n, c, device="cuda") + 0.5, torch.rand(n, c, device="cuda") + 0.5
    noise, bias = torch.randn(n, res * res, device="cuda"), torch.randn(c, device="cuda")
    st = torch.tensor([0.1], device="cuda")
    low = os.environ.get("W3_LOW", "0") == "1" and res >= 256       # the engine's fused skip up-sample (256^2 and larger)
    resid = torch.randn(n, c, res // 2, res // 2, device="cuda") if low else torch.randn(n, c, res, res, device="cuda")
    ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.4, residual=None if low else resid)
    u2 = cv.winograd2_weights(w)
    out = torch.empty_like(x)
    fn = lambda: cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, out=out, residual_low=resid if low else None)
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"pad {os.environ.get('MGF_W3_LDS_PAD','0'):>6} res {res:4d} c {c:3d}: {e0.elapsed_time(e1)/5*1e3:8.1f} us", flush=True)
