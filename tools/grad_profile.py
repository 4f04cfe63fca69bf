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
cfg = GeneratorConfig(img_resolution=1024)
dev = torch.device("cuda", 0)
sd, G, percept, eng, target, latent_mean, latent_std, lms = build(cfg, dev, 0, 64, False, 1)
total = steps + 4
ge = GradientProjectionEngine(Generator(sd, cfg, dev, max_batch=1), target, latent_mean, latent_std, ProjectionArgs(step=total), percept=percept,
                              use_mse=True, lm_target=lms[0], lm_steps=lms[1][:total], noise_mode="random", seed=5, use_graph=False)
ge.run(2)
torch.cuda.synchronize()
ge.run(steps)
torch.cuda.synchronize()
print("done", ge.result()[1:3])
