"""Per-step loss of B = 2 lock-step targets against two single-target gradient engines on the tiny generator (tests/test_hip_gradient.py::
test_gradient_projection_lockstep_targets_equal_single_runs) under a shifted seed: where do the trajectories part?   python tools/soak_lockstep_probe.py SEED"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from morphganformer_amd.engine import Generator
from morphganformer_amd.grad import GeneratorGrad
from morphganformer_amd.lpips import PerceptualLoss
from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs, synthetic_landmarks
from morphganformer_amd.synth_weights import TINY, make_state_dict

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 21
cfg = TINY
G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=2)
GeneratorGrad(G)
steps, B = 8, 2
torch.manual_seed(seed)
latent_mean = torch.randn(cfg.k, cfg.z_dim, device="cuda")
eps = torch.randn(steps, B, cfg.k, cfg.z_dim, device="cuda")
targets = G(torch.randn(B, cfg.k, cfg.z_dim, device="cuda"), None, noise_mode="const")[0].clamp(-1, 1).clone()
lms = [synthetic_landmarks(steps, 64, 9 + j) for j in range(B)]
valid = np.ones((B, steps), np.int32)
valid[1, 2] = 0
for lr in (0.05, 0.005):
    args = ProjectionArgs(step=steps, lr=lr, lr_rampup=0.25)
    singles = []
    for j in range(B):
        e = GradientProjectionEngine(G, targets[j:j + 1].contiguous(), latent_mean, 1.0, args, percept=PerceptualLoss(net="squeeze", allow_random_backbone=True),
                                     lm_target=lms[j][0], lm_steps=lms[j][1], lm_valid=valid[j], eps=eps[:, j:j + 1].contiguous(),
                                     noise_mode="const", use_graph=False).run()
        singles.append(e.result()[3])
    for graph in (True, False):
        multi = GradientProjectionEngine(G, targets, latent_mean, 1.0, args, percept=PerceptualLoss(net="squeeze", allow_random_backbone=True),
                                         lm_target=np.stack([l[0] for l in lms]), lm_steps=np.stack([l[1] for l in lms]), lm_valid=valid,
                                         eps=eps, noise_mode="const", use_graph=graph).run()
        losses = multi.result()[3]
        for j in range(B):
            print(f"lr {lr} graph {graph} target {j}: single", np.array2string(np.asarray(singles[j]), precision=5), " lockstep - single",
                  np.array2string(np.asarray(losses[j]) - np.asarray(singles[j]), precision=2), flush=True)
