"""The stride-2 3x3 convolutions of gradient mode (data gradients of the up-sampling layers) (GPU): python tools/stride2_micro.py [n]   (MGF_TCONV_FIXED=0: run-time geometry)"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for cin, cout, h in ((32, 64, 512), (64, 128, 256), (128, 256, 128), (256, 512, 64)):
    # data gradient of an up layer: stride-2 3x3 conv on the (2h+1)^2 grid -> h^2
    x = torch.randn(n, cin, 2 * h + 1, 2 * h + 1, device="cuda"); s = torch.rand(n, cin, device="cuda") + 0.5
    pc = cv.pack_weights(torch.randn(cout, cin, 3, 3, device="cuda") / math.sqrt(9 * cin))
    out = torch.empty(n, cout, h, h, device="cuda")
    fn = lambda: cv.conv_forward(x, pc, stride=2, pad=(0, 0), in_scale=s, out=out)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{cin:3d}->{cout:3d} out {h:3d}^2 n {n}: {us:7.1f} us  {2*9*cin*cout*h*h*n/us/1e6:6.1f} TF", flush=True)
