"""Run a few gradient-mode projection steps at 1024^2 (for rocprofv3 --kernel-trace); development aid."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build
from morphganformer_amd.engine import Generator
from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs
from morphganformer_amd.synth_weights import GeneratorConfig

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = GeneratorConfig(img_resolution=1024)
dev = torch.device("cuda", 0)
sd, G, percept, eng, target, latent_mean, latent_std, lms = build(cfg, dev, 0, 64, False, 1)
total = steps + 4
import numpy as np
from morphganformer_amd.lpips import PerceptualLoss
GB = Generator(sd, cfg, dev, max_batch=B)
if B > 1:
    target = GB(torch.randn(B, cfg.k, cfg.z_dim, device=dev), None, noise_mode="const")[0].clamp(-1, 1).clone()
    lm_t, lm_s = np.stack([lms[0]] * B), np.stack([lms[1][:total]] * B)
else:
    lm_t, lm_s = lms[0], lms[1][:total]
ge = GradientProjectionEngine(GB, target, latent_mean, latent_std, ProjectionArgs(step=total), percept=PerceptualLoss(net="squeeze", device=dev, allow_random_backbone=True),
                              use_mse=True, lm_target=lm_t, lm_steps=lm_s, noise_mode="random", seed=5, use_graph=False)
ge.run(2)
torch.cuda.synchronize()
ge.run(steps)
torch.cuda.synchronize()
print("done", ge.result()[1])
