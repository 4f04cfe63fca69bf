"""Both loop modes on one synthetic 1024^2 target through drivers.project_image, 400 steps each: the loss trajectory of gradient mode (Adam on the
latent through generator + LPIPS + MSE + Wing) beside literal mode's best-of-N sampling -- a full-size end-to-end sanity run: python tools/grad_long_run.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from morphganformer_amd import drivers
from morphganformer_amd.engine import Generator
from morphganformer_amd.lpips import PerceptualLoss
from morphganformer_amd.projection import ProjectionArgs, synthetic_landmarks
from morphganformer_amd.synth_weights import GeneratorConfig, make_state_dict, synthetic_latents
cfg = GeneratorConfig(img_resolution=1024)
G = Generator(make_state_dict(cfg, seed=0), cfg, "cuda", max_batch=1)
z = torch.from_numpy(synthetic_latents(cfg, 1, seed=1001)).cuda()
target = G(z, None, noise_mode="const")[0].clamp(-1, 1).clone()
P = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, allow_random_backbone=True)
steps = 400
lm_t, lm_s = synthetic_landmarks(steps, 1024, seed=5)
for mode in ("gradient", "literal"):
    t0 = time.time()
    r = drivers.project_image(G, target, lm_t, lm_s, args=ProjectionArgs(step=steps), percept=P, seed=3, mode=mode)
    torch.cuda.synchronize()
    L = np.asarray(r["losses"])
    print(mode, "best", r["loss"], "at", r["step"], "time %.2fs" % (time.time() - t0), "losses[::50]", np.round(L[::50], 4).tolist(), "finite", bool(np.isfinite(L).all()))
