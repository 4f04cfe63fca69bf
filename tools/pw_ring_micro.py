"""Times of the 1x1 GEMM (mgf_conv1x1_f32) on the layer shapes that matter -- generator skips, SqueezeNet Fire squeeze layers, FaceNet's Block35 /
Block17 / Block8 1x1 layers at a 1024^2 input -- for comparing builds of pointwise.hip (MGF_LIB_PATH=exp_build/libmgf_X.so):
    python tools/pw_ring_micro.py [n]      ->  one line per layer: us, TFLOP/s, and the total"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from morphganformer_amd import _lib, conv as cv

LAYERS = [("skip b64", 512, 512, 32, 32), ("skip b128", 512, 256, 64, 64), ("skip b256", 256, 128, 128, 128), ("skip b512", 128, 64, 256, 256),
          ("fire3 sq", 128, 16, 255, 255), ("fire5 sq", 256, 32, 127, 127), ("fire9 sq", 512, 64, 63, 63),
          ("b35 in", 256, 32, 125, 125), ("b35 close", 96, 256, 125, 125), ("b17 in", 896, 128, 62, 62), ("b17 close", 256, 896, 62, 62),
          ("b8 in", 1792, 192, 30, 30), ("b8 close", 384, 1792, 30, 30), ("mixed7a", 896, 256, 62, 62)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    tot = 0.0
    rows = []
    for name, cin, cout, h, w in LAYERS:
        x = torch.randn(n, cin, h, w, device="cuda")
        pc = cv.pack_weights(torch.randn(cout, cin, 1, 1, device="cuda") / math.sqrt(cin))
        out = torch.empty(n, cout, h, w, device="cuda")
        ep = _lib.make_epilogue(bias=torch.randn(cout, device="cuda"), act="relu", gain=1.0)
        fn = lambda: cv.conv_forward(x, pc, epilogue=ep, out=out)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        tot += us
        rows.append(f"{name:<10} {cin:>4}->{cout:<4} {h:>3}x{w:<3} {us:8.1f} us {2.0 * cin * cout * h * w * n / us / 1e6:6.1f} TF")
    print("\n".join(rows))
    print(f"PWTOTAL {os.environ.get('MGF_LIB_PATH', 'product')}: {tot:.1f} us")


if __name__ == "__main__":
    main()
