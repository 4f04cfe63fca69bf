"""Winograd vs direct 3x3 conv on the generator's conv1 layer shapes (development aid): python tools/wino_micro.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from morphganformer_amd import _lib, conv as cv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for res, c in ((16, 512), (32, 512), (64, 512), (128, 256), (256, 128), (512, 64), (1024, 32))[int(os.environ.get("WM_FIRST", "0")):]:
    x = torch.randn(n, c, res, res, device="cuda")
    w = torch.randn(c, c, 3, 3, device="cuda") / (3 * c ** 0.5)
    s, d = torch.rand(n, c, device="cuda") + 0.5, torch.rand(n, c, device="cuda") + 0.5
    noise, bias = torch.randn(n, res * res, device="cuda"), torch.randn(c, device="cuda")
    st = torch.tensor([0.1], device="cuda")
    resid = torch.randn(n, c, res, res, device="cuda") if os.environ.get("WM_RES", "1") != "0" else None     # conv1 layers add the skip branch
    ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.4, residual=resid)
    u, u2, pc = (cv.winograd_weights(w) if c % 64 == 0 else None), cv.winograd2_weights(w), cv.pack_weights(w)
    out_w, out_d, out_2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    fw = (lambda: cv.winograd_forward(x, u, in_scale=s, out_scale=d, epilogue=ep, out=out_w)) if u is not None else (lambda: out_w.copy_(out_d))
    fd = lambda: cv.conv_forward(x, pc, pad=(1, 1), in_scale=s, out_scale=d, epilogue=ep, out=out_d)
    out_3 = torch.empty_like(x)

    def f2():
        cv.WINOGRAD_FORM = 2
        cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, out=out_2)

    def f3():
        cv.WINOGRAD_FORM = 3
        cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, out=out_3)
    res_t = []
    for fn in (fw, fd, f2, f3):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res_t.append(e0.elapsed_time(e1) / 5)
    gf = 2 * 9 * c * c * res * res * n / 1e9
    err = float((out_w - out_d).abs().max() / out_d.abs().max())
    err2 = float((out_2 - out_d).abs().max() / out_d.abs().max())
    print(f"res {res:4d} c {c:4d}: winograd {res_t[0]*1e3:8.1f} us ({gf/res_t[0]:6.1f} TF alg)  direct {res_t[1]*1e3:8.1f} us ({gf/res_t[1]:6.1f} TF)  rel diff {err:.1e}"
          f" | form2 {res_t[2]*1e3:8.1f} us ({gf/res_t[2]:6.1f} TF) diff {err2:.1e} | form3 {res_t[3]*1e3:8.1f} us ({gf/res_t[3]:6.1f} TF) diff "
          f"{float((out_3 - out_d).abs().max() / out_d.abs().max()):.1e}", flush=True)
