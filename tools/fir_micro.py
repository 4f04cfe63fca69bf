"""Micro-benchmark of the wide FIR kernels on the generator's shapes (GPU), with a plain copy of the same bytes beside them.
    python tools/fir_micro.py [res ...]      (MGF_MICRO_N = samples)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from morphganformer_amd import _lib, conv as cv
from tools.conv_micro import N, bench

f1 = torch.tensor([1., 3., 3., 1.])
f2d = (f1[:, None] * f1[None, :] / 64).cuda()
for res in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
    c = min(32768 // res, 512)
    h = res // 2
    t = torch.randn(N, c, 2 * h + 1, cv.tconv_pitch(h), device="cuda")[:, :, :, :2 * h + 1]
    y = torch.empty(N, c, res, res, device="cuda")
    noise = torch.randn(N, res, res, device="cuda")
    ns = torch.ones(1, device="cuda")
    bias = torch.randn(c, device="cuda")
    ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=ns, noise_n=N, act="lrelu", alpha=0.2, gain=1.4)
    us1 = bench(lambda: cv.upfirdn_into(y, t, f2d, up=1, pad=(1, 1, 1, 1), gain=4.0, epilogue=ep, separable=True))
    lo = torch.randn(N, c, h, h, device="cuda")
    us2 = bench(lambda: cv.upfirdn_into(y, lo, f2d, up=2, pad=(2, 1, 2, 1), gain=4.0))
    src = torch.randn(N, c, res, res, device="cuda")
    us3 = bench(lambda: torch.mul(src, 2.0, out=y))
    us4 = bench(lambda: y.fill_(1.0))
    gb = y.numel() * 4 / 1e9
    print(f"r{res} n={N} c={c}: blur+ep {us1:7.1f} us ({2 * gb / us1 * 1e3:5.2f} TB/s)   up2 {us2:7.1f} us ({1.25 * gb / us2 * 1e3:5.2f} TB/s)   "
          f"y=2x {us3:7.1f} us ({2 * gb / us3 * 1e3:5.2f} TB/s)   fill {us4:7.1f} us ({gb / us4 * 1e3:5.2f} TB/s)", flush=True)
