import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from morphganformer_amd import _lib, conv as cv
L = _lib.lib()
"""One-shot form-3 Winograd shapes (11 = 32 channels x 32 tiles, 21 = 64 channels, 12 = 64 tiles) on the generator's conv1 layers: python tools/w3_shape_ab.py [n ...]"""
for n, (c, res) in [(int(a), l) for a in (sys.argv[1:] or ["32"]) for l in ((512, 32), (512, 64), (256, 128), (128, 256), (64, 512))]:
    x = torch.randn(n, c, res, res, device="cuda")
    wt = torch.randn(c, c, 3, 3, device="cuda") / math.sqrt(9 * c)
    u = cv.winograd_pack(wt, 1.0, res)
    s, d = torch.rand(n, c, device="cuda") + 0.5, torch.rand(n, c, device="cuda") + 0.5
    out = torch.empty(n, c, res, res, device="cuda")
    for shape in (11, 21, 12, 11, 21):
        _lib.check(L.mgf_winograd3_force_shape(shape))
        fn = lambda: cv.winograd_forward(x, u, in_scale=s, out_scale=d, out=out)
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"n {n:2d} {c:3d} ch {res:3d}^2 shape {shape}: {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us", flush=True)
    _lib.check(L.mgf_winograd3_force_shape(0))
