"""mgf_lpips_layer_bwd_relu_f32 at the tap sizes of a lockstep-8 LPIPS(squeeze) backward (GPU): python tools/lpips_bwd_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib
L = _lib.lib()
n = 8
for c, s, split in ((64, 511, 64), (128, 255, 64), (256, 127, 128), (384, 63, 192), (512, 63, 256)):
    hw = s * s
    f0 = torch.relu(torch.randn(n, c, hw, device="cuda")); f1 = torch.rand(n, c, hw, device="cuda"); lin = torch.rand(c, device="cuda")
    dy = torch.randn(n, c, hw, device="cuda")
    a, b = torch.empty(n, split, hw, device="cuda"), torch.empty(n, max(c - split, 1), hw, device="cuda")
    f = lambda: _lib.check(L.mgf_lpips_layer_bwd_relu_f32(a.data_ptr(), b.data_ptr() if split < c else None, dy.data_ptr(), f0.data_ptr(),
                                                          f1.data_ptr(), lin.data_ptr(), n, c, split, hw, c * hw, 0.7, _lib.stream_ptr()))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{os.environ.get('MGF_LPIPS_BWD_PXB', 'auto'):<30} c={c:3d} {s}^2: {us:7.1f} us  {4 * f0.numel() * 4 / us / 1e6:5.2f} TB/s (4 tensor passes)", flush=True)
