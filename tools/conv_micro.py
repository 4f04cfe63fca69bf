"""Micro-benchmark of mgf_conv_taps_f32 on the generator's layer shapes (GPU).  MGF_CONV_TILE=wm,wn forces a tile.
    python tools/conv_micro.py [res ...]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from morphganformer_amd import _lib, conv as cv

SHAPES = {  # name: (cin, cout, res, kind)
    "r1024_conv": (32, 32, 1024, "conv"), "r512_conv": (64, 64, 512, "conv"), "r256_conv": (128, 128, 256, "conv"),
    "r128_conv": (256, 256, 128, "conv"), "r64_conv": (512, 512, 64, "conv"), "r32_conv": (512, 512, 32, "conv"),
    "r1024_tconv": (64, 32, 512, "tconv"), "r512_tconv": (128, 64, 256, "tconv"), "r256_tconv": (256, 128, 128, "tconv"),
    "r128_tconv": (512, 256, 64, "tconv"), "r64_tconv": (512, 512, 32, "tconv"), "r32_tconv": (512, 512, 16, "tconv"),
    "r16_tconv": (512, 512, 8, "tconv"), "r8_tconv": (512, 512, 4, "tconv"), "r16_conv": (512, 512, 16, "conv"),
    "r8_conv": (512, 512, 8, "conv"),
    "r1024_skip": (64, 32, 512, "1x1"), "torgb": (32, 3, 1024, "1x1"),
}


N = int(os.environ.get("MGF_MICRO_N", "1"))       # samples per launch


def bench(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    names = sys.argv[1:] or list(SHAPES)
    for name in names:
        cin, cout, res, kind = SHAPES[name]
        x = torch.randn(N, cin, res, res, device="cuda")
        s = 1 + 0.1 * torch.randn(N, cin, device="cuda")
        dsc = 1 + 0.1 * torch.randn(N, cout, device="cuda")
        k = 1 if kind == "1x1" else 3
        w = torch.randn(cout, cin, k, k, device="cuda") / math.sqrt(cin * k * k)
        pc = cv.pack_weights(w)
        if kind == "tconv":
            out = torch.empty(N, cout, 2 * res + 1, cv.tconv_pitch(res), device="cuda")
            fn = lambda: cv.tconv3x3s2_forward(x, pc, in_scale=s, out_scale=dsc, out=out)
            flops = 2 * 9 * cin * cout * res * res * N
        else:
            out = torch.empty(N, cout, res, res, device="cuda")
            fn = lambda: cv.conv_forward(x, pc, pad=(k // 2, k // 2), in_scale=s, out_scale=dsc, out=out)
            flops = 2 * k * k * cin * cout * res * res * N
        us = bench(fn)
        print(f"{name:<14} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  tile={os.environ.get('MGF_CONV_TILE', 'auto')} n={N}", flush=True)


if __name__ == "__main__":
    main()
