"""Biometric-gradient accuracy of the HIP path and of the float32 CPU oracle, both against float64 autograd (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from morphganformer_amd.iresnet import BiometricLoss, IResNetEmbedder, random_state
from oracle.embed_ref import biometric_loss_ref
n, size = 2, 112
sd_np = random_state(18, seed=3)
bio = BiometricLoss(IResNetEmbedder(sd_np, depth=18, n=n, device="cuda"))
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())
for seed in range(1, 7):
    torch.manual_seed(seed)
    pred0 = torch.rand(n, 3, size, size) * 2 - 1
    target = torch.rand(1, 3, size, size) * 2 - 1
    def oracle(dt):
        p = pred0.to(dt).requires_grad_(True)
        sd = {k: torch.from_numpy(v).to(dt) for k, v in sd_np.items()}
        val = biometric_loss_ref(sd, p, target.to(dt).expand(n, -1, -1, -1), 18)
        (g,) = torch.autograd.grad(val.sum() * 0.3, p)
        return val.detach(), g
    v32, g32 = oracle(torch.float32)
    v64, g64 = oracle(torch.float64)
    bio.set_target(target.cuda())
    out = torch.empty(n, device="cuda")
    bio.distance_into(out, pred0.cuda())
    dimg = torch.zeros(n, 3, size, size, device="cuda")
    bio.grad_into(dimg, scale=0.3)
    rms = lambda a: float(((a.double().cpu() - g64).square().mean().sqrt()) / g64.abs().max())
    print(f"seed {seed}: max-rel hip {rel(dimg, g64):.2e} f32 {rel(g32, g64):.2e} | rms hip {rms(dimg):.2e} f32 {rms(g32):.2e}")
