"""Time the bf16x3 conv launches against the float32 ones on the generator's big layer shapes (development aid):
    python tools/bf_micro.py [n=32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import conv as cv          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for kind, cin, cout, res in (("tconv", 64, 32, 512), ("tconv", 128, 64, 256), ("tconv", 256, 128, 128), ("tconv", 512, 256, 64),
                             ("conv3", 32, 32, 1024), ("conv3", 64, 64, 512), ("conv3", 128, 128, 256), ("conv3", 256, 256, 128), ("conv3", 512, 512, 64)):
    x = torch.randn(n, cin, res, res, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    s, dm = torch.rand(n, cin, device="cuda") + 0.5, torch.rand(n, cout, device="cuda") + 0.5
    pc = cv.pack_weights(w)
    wb = cv.pack_weights_bf16x3(pc)
    if kind == "tconv":
        out = torch.empty(n, cout, 2 * res + 1, cv.tconv_pitch(res), device="cuda")
        t32 = bench(lambda: cv.tconv3x3s2_forward(x, pc, in_scale=s, out_scale=dm, out=out))
        tbf = bench(lambda: cv.tconv3x3s2_forward(x, pc, in_scale=s, out_scale=dm, out=out, bf=wb))
        gf = 2 * 9 * cin * cout * res * res * n / 1e9
        extra = ""
    else:
        out = torch.empty(n, cout, res, res, device="cuda")
        t32 = bench(lambda: cv.conv_forward(x, pc, pad=(1, 1), in_scale=s, out_scale=dm, out=out))
        tbf = bench(lambda: cv.conv_forward(x, pc, pad=(1, 1), in_scale=s, out_scale=dm, out=out, bf=wb))
        gf = 2 * 9 * cin * cout * res * res * n / 1e9
        u = cv.winograd_pack(w, 1.0, res)
        tw = bench(lambda: cv.winograd_forward(x, u, in_scale=s, out_scale=dm, out=out))
        extra = f"  winograd f32 {tw:8.1f} us ({gf / tw * 1e-3:6.1f} TF alg)"
    print(f"BFMICRO {kind} n={n} {cin:4d}->{cout:4d} @{res:4d}: f32 taps {t32:8.1f} us ({gf / t32 * 1e-3:6.1f} TF)  bf16x3 {tbf:8.1f} us ({gf / tbf * 1e-3:6.1f} TF)  x{t32 / tbf:.2f}{extra}",
          flush=True)
