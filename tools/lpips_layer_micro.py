"""LPIPS tap-distance kernel on the SqueezeNet tap shapes at 1024^2 (GPU): python tools/lpips_layer_micro.py [n]   (MGF_LPIPS_PXB=16|64 pins the block)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
L = _lib.lib()
scratch = torch.empty(n * 16384 + 16, device="cuda")
out = torch.empty(n, device="cuda")
for c, side in ((128, 255), (256, 127), (384, 63), (384, 63), (512, 63), (512, 63)):
    hw = side * side
    a = torch.randn(n, c, hw, device="cuda"); b = torch.randn(1, c, hw, device="cuda"); lin = torch.rand(c, device="cuda")
    fn = lambda: _lib.check(L.mgf_lpips_layer_f32(out.data_ptr(), a.data_ptr(), b.data_ptr(), lin.data_ptr(), n, c, hw, 0, 0, scratch.data_ptr(), _lib.stream_ptr()))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"c {c:3d} {side}x{side} n {n}: {us:7.1f} us  {a.numel() * 4 / us / 1e6:5.2f} TB/s (candidates' taps)", flush=True)
