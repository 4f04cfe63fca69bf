"""Literal loop with a "no face" step under (graph, pipeline, batch) combinations: the recorded losses and the best step must not depend on them.   python tools/soak_pipeline_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from morphganformer_amd.engine import Generator
from morphganformer_amd.lpips import PerceptualLoss
from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, synthetic_landmarks
from morphganformer_amd.synth_weights import TINY, make_state_dict

cfg = TINY
G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
P = PerceptualLoss(net="squeeze", allow_random_backbone=True)
steps = 5
torch.manual_seed(3)
latent_mean = torch.randn(cfg.k, cfg.z_dim)
eps = torch.randn(steps, 1, cfg.k, cfg.z_dim)
target = G(torch.randn(1, cfg.k, cfg.z_dim).cuda(), None, noise_mode="const")[0].clamp(-1, 1).clone()
lm_t, lm_s = synthetic_landmarks(steps, 64, 51)
for valid in ([1, 1, 1, 1, 1], [1, 1, 0, 1, 1], [0, 1, 1, 1, 1], [1, 1, 1, 1, 0]):
    for use_graph in (False, True):
        for pipeline in (False, True):
            for batch in (1, 2, 3):
                a = ProjectionArgs(step=steps, lamda=0.008, beta=1.1, percept_weight=1.0)
                eng = ProjectionEngine(G, target, latent_mean.cuda(), 1.3, a, percept=P, use_mse=True, eps=eps.cuda(), noise_mode="const", use_graph=use_graph,
                                       batch=batch, pipeline=pipeline, lm_target=lm_t, lm_steps=lm_s, lm_valid=np.array(valid, np.int32))
                try:
                    lat, bstep, bloss, losses = eng.run().result()
                    print(valid, "graph", use_graph, "pipeline", pipeline, "batch", batch, "->", bstep, np.array2string(np.asarray(losses), precision=4), flush=True)
                except IndexError as e:
                    print(valid, "graph", use_graph, "pipeline", pipeline, "batch", batch, "-> IndexError", np.array2string(eng.losses.cpu().numpy() if hasattr(eng, "losses") else np.zeros(1), precision=4), flush=True)
