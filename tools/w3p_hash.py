"""SHA-1 of the persistent form-3 kernel's outputs on fixed seeded inputs (for comparing two builds of wino3.hip bit for bit):
python tools/w3p_hash.py   (MGF_LIB_PATH selects the build)"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib, conv as cv
L = _lib.lib()
torch.manual_seed(0)
h = hashlib.sha1()
for (n, c, co, hh, ww) in ((2, 32, 32, 64, 64), (1, 32, 32, 256, 1024), (3, 32, 64, 32, 512)):
    x = torch.randn(n, c, hh, ww, device="cuda")
    wt = torch.randn(co, c, 3, 3, device="cuda") / (3 * c ** 0.5)
    s, d = torch.rand(n, c, device="cuda") + 0.5, torch.rand(n, co, device="cuda") + 0.5
    noise, bias, st = torch.randn(n, hh * ww, device="cuda"), torch.randn(co, device="cuda"), torch.tensor([0.3], device="cuda")
    low = torch.randn(n, co, hh // 2, ww // 2, device="cuda")
    u2 = cv.winograd2_weights(wt)
    ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.4)
    _lib.check(L.mgf_winograd3_force_shape(31))
    outs = [cv.winograd2_forward(x, u2, in_scale=s, out_scale=d, epilogue=ep, residual_low=low), cv.winograd2_forward(x, u2)]
    if co == 32:
        rw, rb = torch.randn(n, 3, co, device="cuda"), torch.randn(3, device="cuda")
        outs.append(cv.winograd2_rgb_forward(x, u2, rw, rb, torch.empty(n, 3, hh, ww, device="cuda"), in_scale=s, out_scale=d))
    torch.cuda.synchronize()
    _lib.check(L.mgf_winograd3_force_shape(0))
    for o in outs:
        h.update(o.cpu().numpy().tobytes())
print("W3PHASH", os.environ.get("MGF_LIB_PATH", "product"), h.hexdigest())
