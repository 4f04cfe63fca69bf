"""Duplex attention layer on the generator's shapes (GPU): python tools/attn_micro.py [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
L = _lib.lib()
for res, c in ((32, 512), (64, 512), (128, 256)):
    f = res * res
    x = torch.randn(n, c, f, device="cuda"); y = torch.empty_like(x); r = torch.randn_like(x)
    wqc = torch.randn(c, 16, device="cuda") / c ** 0.5; spos = torch.randn(f, 16, device="cuda"); vwb = 1 + 0.1 * torch.randn(n, c, 16, device="cuda")
    noise = torch.randn(n, f, device="cuda"); bias = torch.randn(c, device="cuda"); st = torch.tensor([0.1], device="cuda")
    ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", gain=1.4, residual=r)
    import ctypes as C
    fn = lambda: _lib.check(L.mgf_duplex_attention(y.data_ptr(), x.data_ptr(), wqc.data_ptr(), spos.data_ptr(), vwb.data_ptr(), n, c, f, 16, C.byref(ep), res, None, None, _lib.stream_ptr()))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"res {res:4d} c {c:3d} n {n}: {us:7.1f} us  {3 * x.numel() * 4 / us / 1e6:5.2f} TB/s (x + residual in, y out)", flush=True)
