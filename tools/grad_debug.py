"""Layer-by-layer comparison of the HIP backward pass with autograd through the CPU oracle (development aid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from morphganformer_amd.engine import Generator
from morphganformer_amd.grad import GeneratorGrad
from morphganformer_amd.synth_weights import TINY, make_state_dict
from oracle.generator_ref import mapping_ref, synthesis_ref, to_torch_state

cfg = TINY
sd = make_state_dict(cfg, seed=0)
tsd = to_torch_state(sd)
gg = GeneratorGrad(Generator(sd, cfg, "cuda", max_batch=2))
torch.manual_seed(11)
z = torch.randn(2, cfg.k, cfg.z_dim)
dimg = torch.randn(2, 3, cfg.img_resolution, cfg.img_resolution)
w = mapping_ref(tsd, z, cfg).detach().requires_grad_(True)
taps = {}
img_ref = synthesis_ref(tsd, w, cfg, "const", None, taps)
keys = [k for k in taps if k != "ws" and taps[k].requires_grad]
grads = torch.autograd.grad(img_ref, [w] + [taps[k] for k in keys], dimg)
gref = dict(zip(keys, grads[1:]))
ws = w.detach().cuda().unsqueeze(2).expand(-1, -1, cfg.num_ws, -1)
gg.debug = {}
img = gg.forward(ws=ws, noise_mode="const")
dw = gg.backward_w(dimg.cuda())
rel = lambda a, b: float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max())
for k, v in gg.debug.items():
    name = k.replace(":dc", ":conv").replace(":dout", "")
    if name in gref:
        print(f"{k:40s} rel {rel(v, gref[name]):.3e}")
    else:
        print(f"{k:40s} (no oracle tap)")
print("dw local ", rel(dw[:, :-1], grads[0][:, :-1]), " global ", rel(dw[:, -1], grads[0][:, -1]))
