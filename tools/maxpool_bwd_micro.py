"""Time mgf_maxpool3x3s2_ceil_bwd_f32 at the LPIPS(squeeze) sizes of a lockstep-8 gradient step: python tools/maxpool_bwd_micro.py [n]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from morphganformer_amd import _lib
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for c, s in ((64, 511), (128, 255), (256, 127)):
    o = -(-(s - 3) // 2) + 1
    x = torch.relu(torch.randn(n, c, s, s, device="cuda"))
    dy = torch.randn(n, c, o, o, device="cuda")
    dx = torch.empty_like(x)
    f = lambda: _lib.check(L.mgf_maxpool3x3s2_ceil_bwd_f32(dx.data_ptr(), dy.data_ptr(), x.data_ptr(), n * c, s, s, o, o, _lib.stream_ptr()))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = (2 * x.numel() + dy.numel()) * 4 / 1e9
    print(f"c={c} {s}x{s}: {us:7.1f} us  {gb / us * 1e3:5.2f} TB/s")
