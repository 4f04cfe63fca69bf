"""The LPIPS stem's two narrow convs at lockstep-8 size, streaming kernels vs the MFMA tap-list kernel (GPU): python tools/stem_ends_micro.py [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(n, 3, 1024, 1024, device="cuda")
wt = torch.randn(64, 3, 3, 3, device="cuda") / 5
b = torch.randn(64, device="cuda")
pc = cv.pack_weights(wt); pt = cv.transpose_packed(pc, flip=False)
y = torch.empty(n, 64, 511, 511, device="cuda"); dy = torch.randn_like(y)
g = torch.empty(n, 3, 1023, cv.tconv_pitch(511), device="cuda")
def timed(name, fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:<40} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
for on in (True, False):
    cv.NARROW_CONV = on
    timed(f"stem conv 3->64 (narrow={on})", (lambda: cv.conv3x3s2_few_inputs(x, wt, bias=b, relu=True, out=y)) if on else
          (lambda: cv.conv_forward(x, pc, stride=2, pad=(0, 0), epilogue=_lib.make_epilogue(bias=b, act="relu"), out=y)))
    timed(f"stem data gradient 64->3 (narrow={on})", lambda: cv.tconv3x3s2_forward(dy, pt, out=g))
