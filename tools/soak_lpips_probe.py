"""Is a failing LPIPS-gradient soak case a kink (a ReLU / max-pool unit within rounding of its decision) or a defect?  Re-draws the cases of
tests/test_hip_fuzz.py::test_lpips_gradient_random_non_square_sizes under MGF_FUZZ_OFFSET and prints, per case: HIP vs oracle, and the oracle against
ITSELF after perturbing the input by 1e-6 / 1e-5 (a kink moves the oracle's own gradient by as much as the mismatch):  MGF_FUZZ_OFFSET=2 python tools/soak_lpips_probe.py vgg"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from morphganformer_amd.lpips import PerceptualLoss, WEIGHTS_DIR
from oracle.loss_ref import backbone_random, lpips_ref, squeeze_backbone_random

OFF = int(os.environ.get("MGF_FUZZ_OFFSET", "0")) * 100003
net = sys.argv[1] if len(sys.argv) > 1 else "vgg"


def rel(a, b):
    a = a.detach().double().cpu().numpy(); b = b.detach().double().cpu().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


r = np.random.default_rng({"squeeze": 811, "alex": 812, "vgg": 814}[net] + OFF)
bb = squeeze_backbone_random(0) if net == "squeeze" else backbone_random(net, 0)
lin = np.load(os.path.join(WEIGHTS_DIR, f"lpips_lin_{net}.npz"))
lins = [torch.from_numpy(lin[f"lin{i}"]).float().reshape(-1) for i in range(len(lin.files))]
pl = PerceptualLoss(net=net, allow_random_backbone=True)
lo = {"squeeze": 35, "alex": 70, "vgg": 33}[net]
for case in range(4):
    h, w = int(r.integers(lo, 120)), int(r.integers(lo, 120))
    n = int(r.integers(1, 3))
    torch.manual_seed(9000 + case + OFF)
    pred = (torch.rand(n, 3, h, w) * 2 - 1)
    target = torch.rand(1, 3, h, w) * 2 - 1

    def oracle(p, dt=torch.float32):
        p = p.detach().clone().to(dt).requires_grad_(True)
        bbd = bb if dt == torch.float32 else {k: v.to(dt) for k, v in bb.items()}
        v = lpips_ref(bbd, [l.to(dt) for l in lins], p, target.to(dt).expand(n, -1, -1, -1), net=net)
        (g,) = torch.autograd.grad(v.sum(), p)
        return v.detach(), g

    val, ref = oracle(pred)
    pl.set_target(target.cuda())
    out = torch.empty(n, device="cuda")
    pl.distance_into(out, pred.cuda(), keep_taps=True)
    dimg = torch.zeros(n, 3, h, w, device="cuda")
    pl.grad_into(dimg, scale=1.0)
    torch.manual_seed(1)
    nz = torch.randn_like(pred)
    line = f"case {case} {net} {h}x{w} n={n}: value {rel(out, val.reshape(n)):.2e}  grad HIP vs oracle {rel(dimg, ref):.2e}"
    for eps in (1e-6, 1e-5):
        _, g2 = oracle(pred + eps * nz)
        line += f"  oracle vs oracle(x + {eps:g} noise) {rel(g2, ref):.2e}"
    try:
        _, g64 = oracle(pred, torch.float64)
        line += f"  HIP vs oracle f64 {rel(dimg, g64):.2e}  oracle f32 vs f64 {rel(ref, g64):.2e}"
    except Exception as e:                                   # (the oracle's backbone dict may hold non-tensors)
        line += f"  (f64 oracle: {type(e).__name__})"
    d = (dimg.cpu() - ref).abs().amax(dim=(0, 1))
    ys, xs = torch.nonzero(d > 1e-3 * ref.abs().max(), as_tuple=True)
    if len(ys):
        line += f"  pixels above 1e-3: {len(ys)} in rows {int(ys.min())}..{int(ys.max())}, cols {int(xs.min())}..{int(xs.max())}"
    print(line, flush=True)
