"""Time gradient-mode projection steps at 1024^2 under hipGraph replay (development aid; A/B runs of one knob at a time):
    python tools/grad_time.py [targets=1] [steps=40] [tag]        -> one line: tag, targets, ms per step, iters/s (all targets)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build                                                                  # noqa: E402
from morphganformer_amd.engine import Generator                                         # noqa: E402
from morphganformer_amd.lpips import PerceptualLoss                                     # noqa: E402
from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs     # noqa: E402
from morphganformer_amd.synth_weights import GeneratorConfig                            # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tag = sys.argv[3] if len(sys.argv) > 3 else "-"
cfg = GeneratorConfig(img_resolution=1024)
dev = torch.device("cuda", 0)
sd, G, percept, eng, target, latent_mean, latent_std, lms = build(cfg, dev, 0, 64, False, 1)
del eng
total = 2 * steps + 8
GB = Generator(sd, cfg, dev, max_batch=B)
rng = np.random.Generator(np.random.PCG64(7))
lm_t = rng.integers(256, 768, size=(68, 2)).astype(np.float64)
lm_s = lm_t[None] + rng.integers(-16, 17, size=(total, 68, 2)).astype(np.float64)
if B > 1:
    target = torch.cat([GB(torch.randn(1, cfg.k, cfg.z_dim, device=dev), None, noise_mode="const")[0].clamp(-1, 1) for _ in range(B)]).contiguous()
    lm_t, lm_s = np.stack([lm_t] * B), np.stack([lm_s] * B)
ge = GradientProjectionEngine(GB, target, latent_mean, latent_std, ProjectionArgs(step=total), percept=PerceptualLoss(net="squeeze", device=dev, allow_random_backbone=True),
                              use_mse=True, lm_target=lm_t, lm_steps=lm_s, noise_mode="random", seed=5, use_graph=True)
ge.run(4)
torch.cuda.synchronize()
best = 1e9
for _ in range(2):
    t0 = time.perf_counter()
    ge.run(steps)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps)
print(f"GRADTIME {tag} targets={B} ms_per_step={best * 1e3:.3f} iters_per_s={B / best:.2f}", flush=True)
