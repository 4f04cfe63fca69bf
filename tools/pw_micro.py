"""Micro-benchmark of the 1x1 layers of one projection iteration (GPU): mgf_conv1x1_f32 (1 / 2 channel blocks per wave) next to the
tap-list kernel, with the HBM and matrix-pipe bounds of each layer.   python tools/pw_micro.py [n]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from morphganformer_amd import _lib, conv as cv

# (name, cin, cout, h, w, ctotal): the resnet skips of the synthesis blocks, the Fire squeeze / expand1x1 layers of SqueezeNet1.1 at 1024^2
LAYERS = [("skip b8", 512, 512, 4, 4, 512), ("skip b16", 512, 512, 8, 8, 512), ("skip b32", 512, 512, 16, 16, 512),
          ("skip b64", 512, 512, 32, 32, 512), ("skip b128", 512, 256, 64, 64, 256),
          ("skip b256", 256, 128, 128, 128, 128), ("skip b512", 128, 64, 256, 256, 64), ("skip b1024", 64, 32, 512, 512, 32),
          ("fire2 sq", 64, 16, 255, 255, 16), ("fire2 e1", 16, 64, 255, 255, 128), ("fire3 sq", 128, 16, 255, 255, 16),
          ("fire4 sq", 128, 32, 127, 127, 32), ("fire4 e1", 32, 128, 127, 127, 256), ("fire5 sq", 256, 32, 127, 127, 32),
          ("fire6 sq", 256, 48, 63, 63, 48), ("fire6 e1", 48, 192, 63, 63, 384), ("fire7 sq", 384, 48, 63, 63, 48),
          ("fire8 sq", 384, 64, 63, 63, 64), ("fire8 e1", 64, 256, 63, 63, 512), ("fire9 sq", 512, 64, 63, 63, 64)]


def bench(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    L = _lib.lib()
    tot = [0.0, 0.0, 0.0, 0.0]
    for name, cin, cout, h, w, ctotal in LAYERS:
        x = torch.randn(n, cin, h, w, device="cuda")
        pc = cv.pack_weights(torch.randn(cout, cin, 1, 1, device="cuda") / math.sqrt(cin))
        b = torch.randn(cout, device="cuda")
        out = torch.empty(n, ctotal, h, w, device="cuda")
        ep = _lib.make_epilogue(bias=b, act="relu")
        fn = lambda: cv.conv_forward(x, pc, epilogue=ep, out=out)
        us = []
        for mode in ("taps", 1, 2):
            cv.POINTWISE = mode != "taps"
            if mode != "taps":
                _lib.check(L.mgf_conv1x1_force_shape(mode))
            us.append(bench(fn))
        _lib.check(L.mgf_conv1x1_force_shape(0))
        us.append(bench(fn))
        flops = 2.0 * cin * cout * h * w * n
        byts = 4.0 * n * h * w * (cin + cout)
        bound = max(flops / 157.3e12, byts / 5.0e12) * 1e6
        for i in range(4):
            tot[i] += us[i]
        if os.environ.get("PW_MICRO_SHORT"):
            print(f"{name:<11} taps {us[0]:6.1f} cb1 {us[1]:6.1f} cb2 {us[2]:6.1f} auto {us[3]:6.1f} bound {bound:6.1f}", flush=True)
            continue
        print(f"{name:<11} {cin:>3}->{cout:<3} {h:>3}x{w:<3}  taps {us[0]:7.1f}  cb1 {us[1]:7.1f}  cb2 {us[2]:7.1f}  auto {us[3]:7.1f} us   "
              f"bound {bound:6.1f} us ({'mfma' if flops / 157.3e12 > byts / 5.0e12 else 'hbm @5 TB/s'})  auto: {flops / us[3] / 1e6:6.1f} TF {byts / us[3] / 1e6:6.2f} TB/s", flush=True)
    print(f"total       taps {tot[0]:.0f}  cb1 {tot[1]:.0f}  cb2 {tot[2]:.0f}  auto {tot[3]:.0f} us")


if __name__ == "__main__":
    main()
