"""The generator's conv1 layers on the ONE-SHOT form-3 Winograd kernel at 32 samples, with the epilogues the literal loop gives them (noise / bias /
lrelu; attention layers take none: their epilogue runs in the attention kernel; the 256^2 / 512^2 layers up-sample the half-resolution skip) -- for
ablation builds (tools/patches/w3_oneshot_ablation.patch, tools/build_exp.sh w3oK "-DW3O_ABL=K" wino3.hip): python tools/w3_layers_micro.py [n]"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for c, res, att in ((512, 32, True), (512, 64, True), (256, 128, True), (128, 256, False), (64, 512, False)):
    x = torch.randn(n, c, res, res, device="cuda")
    wt = torch.randn(c, c, 3, 3, device="cuda") / math.sqrt(9 * c)
    u = cv.winograd_pack(wt, 1.0, res)
    s, d = torch.rand(n, c, device="cuda") + 0.5, torch.rand(n, c, device="cuda") + 0.5
    out = torch.empty(n, c, res, res, device="cuda")
    ep, low = None, None
    if not att:
        noise, bias, st = torch.randn(n, res * res, device="cuda"), torch.randn(c, device="cuda"), torch.tensor([0.1], device="cuda")
        ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.0)
        low = torch.randn(n, c, res // 2, res // 2, device="cuda")
    fn = lambda: cv.winograd_forward(x, u, in_scale=s, out_scale=d, epilogue=ep, out=out, residual_low=low)
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    floor = 2 * 9 * c * c * res * res * n * (4 / 9) / 157.3e12 * 1e6
    print(f"{os.environ.get('MGF_LIB_PATH', 'default')[-22:]:<22} {c:3d} ch at {res:3d}^2 n {n}: {us:7.1f} us  (matrix work at the 2.4 GHz peak {floor:6.1f} us: {floor / us:.2f})", flush=True)
