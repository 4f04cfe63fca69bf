"""Per-phase shader-clock breakdown of the persistent conv kernel (needs the MGF_EXP=3 experiment build of the library).
    MGF_LIB_PATH=exp_build/libmgf_exp3.so python tools/conv_phases.py r1024_conv ..."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from morphganformer_amd import conv as cv
from tools.conv_micro import N, SHAPES

for name in sys.argv[1:]:
    cin, cout, res, kind = SHAPES[name]
    x = torch.randn(N, cin, res, res, device="cuda")
    s = 1 + 0.1 * torch.randn(N, cin, device="cuda")
    dsc = 1 + 0.1 * torch.randn(N, cout, device="cuda")
    k = 1 if kind == "1x1" else 3
    pc = cv.pack_weights(torch.randn(cout, cin, k, k, device="cuda") / math.sqrt(cin * k * k))
    ws = cv._workspace(0)
    for _ in range(3):
        ws.zero_()
        if kind == "tconv":
            cv.tconv3x3s2_forward(x, pc, in_scale=s, out_scale=dsc)
        else:
            cv.conv_forward(x, pc, pad=(k // 2, k // 2), in_scale=s, out_scale=dsc)
        torch.cuda.synchronize()
    dbg = ws.view(torch.int64)[:1024 * 8].view(1024, 8).cpu()
    dbg = dbg[dbg[:, 4] > 0].double()
    t = dbg[:, 4].sum()
    print(f"{name}: {len(dbg)} workgroups, {int(t)} tiles; cycles per tile: prologue {dbg[:,0].sum()/t:.0f}  chunk-loop {dbg[:,1].sum()/t:.0f} "
          f"(of which mfma phases {dbg[:,3].sum()/t:.0f})  epilogue {dbg[:,2].sum()/t:.0f}")
