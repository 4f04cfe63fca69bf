"""mgf_maxpool3x3s2_ceil_f32 at the LPIPS(squeeze) sizes of the literal loop (25 candidates): python tools/maxpool_fwd_micro.py [n]   (MGF_POOL_TILED=0: row-per-wave form)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for c, s in ((64, 511), (128, 255), (256, 127)):
    o = -(-(s - 3) // 2) + 1
    x = torch.randn(n, c, s, s, device="cuda"); y = torch.empty(n, c, o, o, device="cuda")
    f = lambda: _lib.check(L.mgf_maxpool3x3s2_ceil_f32(y.data_ptr(), x.data_ptr(), n * c, s, s, o, o, _lib.stream_ptr()))
    f(); torch.cuda.synchronize()
    ok = torch.equal(y[:2], torch.nn.functional.max_pool2d(x[:2], 3, 2, ceil_mode=True))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"tiled={os.environ.get('MGF_POOL_TILED', '1')} c={c} {s}^2: {us:7.1f} us  {(x.numel() + y.numel()) * 4 / us / 1e6:5.2f} TB/s  exact={ok}", flush=True)
