"""Border kernel of the transposed conv on the generator's up-layers (GPU): python tools/border_micro.py"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
n = 25
L = _lib.lib()
for cin, cout, h in ((512, 512, 32), (512, 256, 64), (256, 128, 128), (128, 64, 256), (64, 32, 512)):
    x = torch.randn(n, cin, h, h, device="cuda"); s = torch.rand(n, cin, device="cuda") + 0.5; d = torch.rand(n, cout, device="cuda") + 0.5
    pc = cv.pack_weights(torch.randn(cout, cin, 3, 3, device="cuda") / math.sqrt(9 * cin))
    pitch = cv.tconv_pitch(h); oh = 2 * h + 1
    t = torch.zeros(n, cout, oh, pitch, device="cuda")
    fn = lambda: _lib.check(L.mgf_tconv3x3s2_border_f32(t.data_ptr(), x.data_ptr(), pc.wp.data_ptr(), s.data_ptr(), d.data_ptr(), n, cin, h, h, cout, pc.cout_pad,
                                                         pitch, oh * pitch, cout * oh * pitch, cout, _lib.stream_ptr()))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{cin:3d}->{cout:3d} {h:3d}^2: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us", flush=True)
