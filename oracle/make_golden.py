"""Generate tests/golden/*.npz by running the REFERENCE itself (CPU, build container only).

Run:  python -m oracle.make_golden            (from the repo root; needs /root/reference)

The reference cannot travel to the GPU box, so only the small input/output vectors emitted here are
committed.  Harness-side accommodations, none of which alter reference arithmetic:
  * `termcolor` / `seaborn` are absent cosmetic imports of misc.py -> stub modules;
  * `TransformerLayer.dim` is read but its assignment is commented out (networks.py:581,616,814)
    -> property returning to_queries.weight.shape[0].
Weights are the build's own seeded synthetic tensors (morphganformer_amd/synth_weights.py) loaded through
`load_state_dict`, since no checkpoint exists offline.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.path.insert(0, REF)
    for name, attrs in (("termcolor", {"colored": lambda s, *a, **k: s}),
                        ("seaborn", {"color_palette": lambda *a, **k: []})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    from training import networks
    from torch_utils.ops import bias_act, upfirdn2d, conv2d_resample
    import wing_loss
    import adaptive_wing_loss
    networks.TransformerLayer.dim = property(lambda s: s.to_queries.weight.shape[0])
    return types.SimpleNamespace(networks=networks, bias_act=bias_act, upfirdn2d=upfirdn2d,
                                 conv2d_resample=conv2d_resample, wing_loss=wing_loss,
                                 adaptive_wing_loss=adaptive_wing_loss)


def build_reference_generator(ref, cfg, sd_np):
    G = ref.networks.Generator(
        z_dim=cfg.z_dim, c_dim=0, w_dim=cfg.w_dim, k=cfg.k, img_resolution=cfg.img_resolution, img_channels=3,
        mapping_kwargs=dict(num_heads=1, transformer=True, use_pos=True, resnet=True, ltnt2ltnt=True,
                            normalize_global=cfg.normalize_global),
        synthesis_kwargs=dict(channel_base=cfg.channel_base, channel_max=cfg.channel_max, architecture="resnet",
                              style=True, local_noise=True, transformer=True, use_pos=True, num_heads=1,
                              start_res=0, end_res=cfg.attn_max_log2res, norm="layer", integration="mul",
                              kmeans=True, kmeans_iters=1, pos_type="sinus", pos_init="uniform",
                              pos_directions_num=2)).eval().requires_grad_(False)
    state = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    missing, unexpected = G.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    return G


def gold_bias_act(ref):
    from oracle.ops_ref import ACT_NAMES
    rng = np.random.Generator(np.random.PCG64(11))
    out = {}
    x = rng.standard_normal((2, 5, 7, 3)).astype(np.float32) * 2
    b = rng.standard_normal(5).astype(np.float32)
    dy = rng.standard_normal((2, 5, 7, 3)).astype(np.float32)
    ddx = rng.standard_normal((2, 5, 7, 3)).astype(np.float32)
    out["x"], out["b"], out["dy"], out["ddx"] = x, b, dy, ddx
    for act in ACT_NAMES:
        for clamp in (None, 0.5):
            tag = f"{act}_c{'none' if clamp is None else clamp}"
            xt = torch.from_numpy(x).requires_grad_(True)
            y = ref.bias_act.bias_act(xt, torch.from_numpy(b), dim=1, act=act, clamp=clamp, impl="ref")
            (dx,) = torch.autograd.grad(y, xt, torch.from_numpy(dy), create_graph=True)
            out[f"y_{tag}"] = y.detach().numpy()
            out[f"dx_{tag}"] = dx.detach().numpy()
            if dx.requires_grad:
                (d2,) = torch.autograd.grad(dx, xt, torch.from_numpy(ddx), allow_unused=True)
                out[f"d2_{tag}"] = (torch.zeros_like(xt) if d2 is None else d2).detach().numpy()
            else:
                out[f"d2_{tag}"] = np.zeros_like(x)
    # dim=0 bias on a 2-D tensor, and an explicit alpha/gain
    x2 = rng.standard_normal((3, 7)).astype(np.float32)
    b2 = rng.standard_normal(3).astype(np.float32)
    out["x2"], out["b2"] = x2, b2
    out["y2"] = ref.bias_act.bias_act(torch.from_numpy(x2), torch.from_numpy(b2), dim=0, act="lrelu", alpha=0.3,
                                      gain=1.7, impl="ref").numpy()
    np.savez_compressed(os.path.join(OUT, "ops_bias_act.npz"), **out)


UPFIRDN_CASES = [
    # name, shape, filter taps (1-D list => outer product, like the reference for <8 taps), up, down, padding, gain, flip
    ("up2_4x4_g4", (1, 3, 9, 9), [1, 3, 3, 1], 2, 1, [2, 1, 2, 1], 4.0, False),
    ("up1_4x4_crop", (2, 3, 11, 11), [1, 3, 3, 1], 1, 1, [1, 1, 1, 1], 4.0, False),
    ("up2_2x2_nn", (1, 4, 6, 6), [1, 1], 2, 1, [1, 0, 1, 0], 4.0, False),
    ("down2_4x4", (1, 3, 12, 12), [1, 3, 3, 1], 1, 2, [1, 1, 1, 1], 1.0, False),
    ("asym_3x5_flip", (1, 2, 8, 10), None, 1, 1, [2, 1, 0, 3], 1.0, True),
    ("neg_pad", (1, 2, 10, 10), [1, 2, 1], 1, 1, [-1, -2, 0, -1], 1.0, False),
    ("up3_down2", (1, 2, 7, 5), [1, 4, 6, 4, 1], 3, 2, [3, 2, 1, 4], 2.0, False),
    ("sep8", (1, 2, 9, 9), [1, 2, 3, 4, 4, 3, 2, 1], 2, 1, [4, 3, 4, 3], 4.0, False),
]


def gold_upfirdn2d(ref):
    rng = np.random.Generator(np.random.PCG64(12))
    out = {}
    for name, shape, taps, up, down, pad, gain, flip in UPFIRDN_CASES:
        x = rng.standard_normal(shape).astype(np.float32)
        if taps is None:
            f = torch.from_numpy(rng.standard_normal((3, 5)).astype(np.float32))
        else:
            f = ref.upfirdn2d.setup_filter(taps)
        y = ref.upfirdn2d.upfirdn2d(torch.from_numpy(x), f, up=up, down=down, padding=pad, flip_filter=flip,
                                    gain=gain, impl="ref")
        out[f"x_{name}"], out[f"f_{name}"], out[f"y_{name}"] = x, f.numpy(), y.numpy()
    np.savez_compressed(os.path.join(OUT, "ops_upfirdn2d.npz"), **out)


def gold_modconv(ref):
    rng = np.random.Generator(np.random.PCG64(13))
    out = {}
    f = ref.upfirdn2d.setup_filter([1, 3, 3, 1])
    x = rng.standard_normal((2, 8, 8, 8)).astype(np.float32)
    w = (rng.standard_normal((6, 8, 3, 3)) / np.sqrt(72)).astype(np.float32)
    s = (1 + 0.3 * rng.standard_normal((2, 8))).astype(np.float32)
    out["x"], out["w"], out["s"], out["f"] = x, w, s, f.numpy()
    for up in (1, 2):
        for demod in (True, False):
            y = ref.networks.modulated_conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(s), up=up,
                                              padding=1, resample_kernel=f, demodulate=demod, flip_weight=(up == 1),
                                              fused_modconv=True)
            out[f"y_up{up}_demod{int(demod)}"] = y.numpy()
    w1 = (rng.standard_normal((5, 8, 1, 1)) / np.sqrt(8)).astype(np.float32)
    out["w1"] = w1
    out["y_skip_up2"] = ref.conv2d_resample.conv2d_resample(torch.from_numpy(x), torch.from_numpy(w1), f=f, up=2,
                                                            padding=0, flip_weight=False).numpy()
    np.savez_compressed(os.path.join(OUT, "ops_modconv.npz"), **out)


def gold_generator_tiny(ref):
    from morphganformer_amd.synth_weights import TINY, make_state_dict, synthetic_latents
    sd = make_state_dict(TINY, seed=0)
    G = build_reference_generator(ref, TINY, sd)
    z = torch.from_numpy(synthetic_latents(TINY, 2, seed=1000))
    out = {"z": z.numpy()}
    taps = {}
    hooks = []
    for res in TINY.block_resolutions:
        blk = getattr(G.synthesis, f"b{res}")
        hooks.append(blk.register_forward_hook(lambda m, i, o, r=res: taps.__setitem__(f"b{r}", o[0].detach().clone())))
    probs = {}
    for res, name in ((4, "conv1"), (16, "conv0"), (64, "conv1")):
        tr = getattr(getattr(G.synthesis, f"b{res}"), name).transformer
        hooks.append(tr.register_forward_hook(lambda m, i, o, k=f"b{res}.{name}": probs.__setitem__(k, o[1].detach().clone())))
    img, ws = G(z, None, noise_mode="const", return_ws=True)
    for h in hooks:
        h.remove()
    out["img_const"] = img.numpy()
    out["ws"] = ws[:, :, 0].numpy()
    assert float((ws - ws[:, :, :1]).abs().max()) == 0.0
    for k_, v in taps.items():
        out[f"tap_{k_}"] = v.numpy()
    for k_, v in probs.items():
        out[f"probs_{k_}"] = v.reshape(v.shape[0], -1, TINY.k - 1).numpy()
    out["img_none"] = G(z, None, noise_mode="none")[0].numpy()
    # injected noise: patch torch.randn inside the reference module namespace with a deterministic queue
    rng = np.random.Generator(np.random.PCG64(77))
    inj = {}
    order = []
    for res in TINY.block_resolutions:
        for name in (["conv0"] if res > 4 else []) + ["conv1"]:
            key = f"synthesis.b{res}.{name}"
            inj[key] = rng.standard_normal((2, res, res)).astype(np.float32)
            order.append(key)
    queue = list(order)
    real_randn = torch.randn

    def fake_randn(shape, *a, **k):
        key = queue.pop(0)
        t = torch.from_numpy(inj[key]).reshape(shape)
        return t

    ref.networks.torch.randn = fake_randn
    try:
        out["img_inject"] = G(z, None, noise_mode="random")[0].numpy()
    finally:
        ref.networks.torch.randn = real_randn
    assert not queue
    for key, v in inj.items():
        out["noise_" + key] = v
    # gradient-mode oracle: d(mean(img^2))/dz through the reference module
    zg = z.clone().requires_grad_(True)
    G.requires_grad_(False)
    loss = G(zg, None, noise_mode="const")[0].square().mean()
    (gz,) = torch.autograd.grad(loss, zg)
    out["loss_sq"] = np.float32(loss.item())
    out["grad_z"] = gz.numpy()
    np.savez_compressed(os.path.join(OUT, "gen_tiny.npz"), **out)
    return G, sd


def gold_loop_tiny(ref, G):
    """Literal-mode mini run: 50 steps, MSE + lamda*Wing on injected landmarks."""
    from morphganformer_amd.synth_weights import TINY, synthetic_latents
    from oracle.loss_ref import latent_stats_ref, noise_strength_ref
    torch.manual_seed(0)
    steps = 50
    rng = np.random.Generator(np.random.PCG64(5))
    samples = torch.from_numpy(rng.standard_normal((1000, TINY.k, TINY.z_dim)).astype(np.float32))
    latent_mean = samples.mean(0)
    latent_std = ((samples - latent_mean).pow(2).sum() / samples.shape[0]) ** 0.5
    target = G(torch.from_numpy(synthetic_latents(TINY, 1, seed=1001)), None, noise_mode="const")[0].clamp(-1, 1)
    eps = rng.standard_normal((steps, 1, TINY.k, TINY.z_dim)).astype(np.float32)
    lm_target = rng.integers(8, 56, size=(68, 2)).astype(np.float64)
    lm_steps = lm_target[None] + rng.integers(-6, 7, size=(steps, 68, 2)).astype(np.float64)
    wing = ref.wing_loss.WingLoss()
    mse = torch.nn.MSELoss()
    latent_in = latent_mean[None].clone()
    min_loss, best, best_step = 100.0, None, -1
    losses = np.zeros(steps, np.float64)
    sigmas, latents_n = [], []
    for i in range(steps):
        t = i / steps
        sigma = latent_std * 0.05 * max(0, 1 - t / 0.75) ** 2
        latent_n = latent_in + torch.from_numpy(eps[i]) * sigma.item()
        sigmas.append(sigma.item()); latents_n.append(latent_n.numpy().copy())
        img = G(latent_n, 0.7, noise_mode="const")[0]
        w = wing(torch.from_numpy(lm_steps[i]), torch.from_numpy(lm_target))
        total = 0.01 * w + 1.0 * mse(img, target)
        losses[i] = float(total)
        if float(total) < min_loss:
            min_loss, best, best_step = float(total), latent_n.clone(), i
    np.savez_compressed(os.path.join(OUT, "loop_tiny.npz"), latent_mean=latent_mean.numpy(),
                        latent_std=np.float32(latent_std.item()), target=target.numpy(), eps=eps,
                        lm_target=lm_target, lm_steps=lm_steps, losses=losses, best_latent=best.numpy(),
                        best_step=np.int64(best_step), best_loss=np.float64(min_loss), sigmas=np.array(sigmas, np.float64),
                        latents_n=np.stack(latents_n))


def gold_loop_copies_tiny(ref, G):
    """projection_example_v2_percept.py:131-203 on the tiny generator: the latent as 18 noisy copies averaged by torch.mean before the
    generator (the script's own tensor statements, :133-140,154-159, on its [1, 18, k D] layout), min_loss starting at 1.0 (:146), best
    latent = that mean (:193).  The script scores LPIPS(vgg) alone; the fixture scores 0.01 x the pixel MSE against a near target so that it needs
    no backbone -- what it pins is the copies' arithmetic, the averaged latents of every step and the selection."""
    from morphganformer_amd.synth_weights import TINY, synthetic_latents
    steps, n_latent = 24, 18
    rng = np.random.Generator(np.random.PCG64(1818))
    numel = TINY.k * TINY.z_dim
    samples = torch.from_numpy(rng.standard_normal((1000, numel)).astype(np.float32))
    latent_mean = samples.mean(0)                                                       # [k D]: the script flattens the latent (:244-249)
    latent_std = ((samples - latent_mean).pow(2).sum() / samples.shape[0]) ** 0.5
    # (a target near the start latent: the script's min_loss starts at 1.0, far below the pixel MSE of an unrelated image)
    z_t = latent_mean.reshape(1, TINY.k, TINY.z_dim) + torch.from_numpy(rng.standard_normal((1, TINY.k, TINY.z_dim)).astype(np.float32)) * 0.1
    target = G(z_t, None, noise_mode="const")[0].clamp(-1, 1)
    eps = rng.standard_normal((steps, 1, n_latent, numel)).astype(np.float32)
    mse = torch.nn.MSELoss()
    latent_in = latent_mean.detach().clone().unsqueeze(0).repeat(1, 1)                  # :133
    latent_in = latent_in.unsqueeze(1).repeat(1, n_latent, 1)                           # :140
    min_loss, best, best_step = 1.0, None, -1
    losses = np.zeros(steps, np.float64)
    ims = []
    for i in range(steps):
        t = i / steps
        noise_strength = latent_std * 0.05 * max(0, 1 - t / 0.75) ** 2
        latent_n = latent_in + torch.from_numpy(eps[i]) * noise_strength.item()         # latent_noise, :70-72 with the draw injected
        im_latent = torch.mean(latent_n, 1)
        im_latent = im_latent.reshape([1, TINY.k, TINY.z_dim])
        ims.append(im_latent.numpy().copy())
        img = G(im_latent, 0.7, noise_mode="const")[0]
        total = 0.01 * mse(img, target)          # (a coefficient like the drivers' beta: the tiny generator's images are not in [-1, 1] at this latent)
        losses[i] = float(total)
        if float(total) < min_loss:
            min_loss, best, best_step = float(total), im_latent.clone(), i
    assert best_step >= 0
    np.savez_compressed(os.path.join(OUT, "loop_copies_tiny.npz"), latent_mean=latent_mean.numpy(), latent_std=np.float32(latent_std.item()),
                        target=target.numpy(), eps=eps, losses=losses, best_latent=best.numpy(), best_step=np.int64(best_step),
                        best_loss=np.float64(min_loss), im_latents=np.stack(ims), copies=np.int64(n_latent))


def _attention_layer_names(cfg):
    """The synthesis layers that carry a TransformerLayer, in execution order (conv0 before conv1 inside a block)."""
    names = []
    for res in cfg.block_resolutions:
        if cfg.has_attention(res):
            names += ([f"b{res}.conv0"] if res > 4 else []) + [f"b{res}.conv1"]
    return names


def gold_generator_full(ref):
    """Full-size 1024^2 generator: 4096 sampled pixels + per-block checksums -> gen_full1024.npz, and the
    integer gate of SURVEY 8d at full size -> att_full1024.npz: for each of the 11 TransformerLayers of the 1024^2 model
    (networks.py:505-524,776-792) the per-pixel argmax latent assignment (uint8), the mask of pixels whose top-2 probability margin
    exceeds 1e-4 ("decided": an f32 kernel cannot be asked to reproduce a tie) and 256 sampled probability rows, for two latents."""
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    sd = make_state_dict(FULL1024, seed=0)
    G = build_reference_generator(ref, FULL1024, sd)
    # the projection loop discards list2tensor's 738 MB output; skip it for speed
    G.synthesis.list2tensor = lambda att_list, device: torch.zeros([1])
    z = torch.from_numpy(synthetic_latents(FULL1024, 1, seed=1000))
    stats = {}
    probs = {}
    hooks = []
    for res in FULL1024.block_resolutions:
        blk = getattr(G.synthesis, f"b{res}")
        hooks.append(blk.register_forward_hook(
            lambda m, i, o, r=res: stats.__setitem__(r, (float(o[0].double().mean()), float(o[0].double().square().mean().sqrt())))))
    att_names = _attention_layer_names(FULL1024)
    for key in att_names:
        b, name = key.split(".")
        tr = getattr(getattr(G.synthesis, b), name).transformer
        hooks.append(tr.register_forward_hook(lambda m, i, o, k=key: probs.__setitem__(k, o[1].detach().clone())))
    torch.set_num_threads(8)
    img = G(z, None, noise_mode="const")[0]
    T = FULL1024.k - 1
    att = {"z": [z.numpy()], "layers": np.array(att_names)}
    rng_att = np.random.Generator(np.random.PCG64(515))
    rows = {key: np.sort(rng_att.choice(probs[key].numel() // T, size=min(256, probs[key].numel() // T), replace=False)) for key in att_names}

    def keep_att(slot):
        for key in att_names:
            p = probs[key].reshape(-1, T).numpy()                       # [F, T] of the one sample
            top2 = np.sort(p, axis=-1)[:, -2:]
            att.setdefault("argmax_" + key, []).append(p.argmax(-1).astype(np.uint8))
            att.setdefault("decided_" + key, []).append(np.packbits((top2[:, 1] - top2[:, 0]) > 1e-4))
            att.setdefault("probs_" + key, []).append(p[rows[key]])
    keep_att(0)
    for h in hooks[:len(FULL1024.block_resolutions)]:
        h.remove()
    z2 = torch.from_numpy(synthetic_latents(FULL1024, 1, seed=1001))
    att["z"].append(z2.numpy())
    G(z2, None, noise_mode="const")
    keep_att(1)
    for h in hooks:
        h.remove()
    for key in att_names:
        att["rows_" + key] = rows[key]
    np.savez_compressed(os.path.join(OUT, "att_full1024.npz"), **{k_: (np.concatenate(v) if k_ == "z" else np.stack(v) if isinstance(v, list) else v)
                                                                   for k_, v in att.items()})
    rng = np.random.Generator(np.random.PCG64(99))
    idx = rng.integers(0, 3 * 1024 * 1024, size=4096)
    flat = img.reshape(-1).numpy()
    np.savez_compressed(os.path.join(OUT, "gen_full1024.npz"), z=z.numpy(), idx=idx, pixels=flat[idx],
                        img_mean=np.float64(img.double().mean()), img_rms=np.float64(img.double().square().mean().sqrt()),
                        img_absmax=np.float32(img.abs().max()),
                        block_res=np.array(FULL1024.block_resolutions),
                        block_mean=np.array([stats[r][0] for r in FULL1024.block_resolutions]),
                        block_rms=np.array([stats[r][1] for r in FULL1024.block_resolutions]),
                        img_ds=torch.nn.functional.avg_pool2d(img, 16).numpy())


def gold_att_tiny(ref):
    """list2tensor fixture: the stacked attention-map tensor of G(z, return_att=True) on the tiny generator, sub-sampled 8x."""
    from morphganformer_amd.synth_weights import TINY, make_state_dict, synthetic_latents
    G = build_reference_generator(ref, TINY, make_state_dict(TINY, seed=0))
    z = torch.from_numpy(synthetic_latents(TINY, 2, seed=1000))
    img, att = G(z, None, noise_mode="const", return_att=True)
    np.savez_compressed(os.path.join(OUT, "att_tiny.npz"), z=z.numpy(), shape=np.array(att.shape), att_sub=att[:, :, :, 0, 3::8, 5::8].numpy())
    print("att_tiny", tuple(att.shape))


def gold_wplus_tiny(ref):
    """Boundary fixture for Generator.forward's less-travelled arguments (networks.py:1304-1331) on the tiny generator:
    distinct per-layer latents through `ws=`, truncation with a cutoff (mapping, :935-941), and the return_att tensor."""
    from morphganformer_amd.synth_weights import TINY, make_state_dict, synthetic_latents
    G = build_reference_generator(ref, TINY, make_state_dict(TINY, seed=0))
    z = torch.from_numpy(synthetic_latents(TINY, 2, seed=1000))
    rng = np.random.Generator(np.random.PCG64(4242))
    ws0 = G(z, None, noise_mode="const", subnet="mapping")
    ws = (ws0 + torch.from_numpy(rng.standard_normal(tuple(ws0.shape)).astype(np.float32)) * 0.5 * ws0.std()).contiguous()
    assert float((ws - ws[:, :, :1]).abs().max()) > 0
    img_ws, att_ws = G(ws=ws, noise_mode="const", return_att=True)
    img_cut, ws_cut = G(z, None, truncation_psi=0.6, truncation_cutoff=5, noise_mode="const", return_ws=True)
    img_psi, ws_psi = G(z, None, truncation_psi=0.6, noise_mode="const", return_ws=True)
    # gradient-mode oracle for W+: d(mean(img^2)) / d(ws) through the REFERENCE module's autograd, every layer slot its own gradient
    wg = ws.clone().requires_grad_(True)
    G.requires_grad_(False)
    loss_ws = G(ws=wg, noise_mode="const")[0].square().mean()
    (grad_ws,) = torch.autograd.grad(loss_ws, wg)
    assert float(grad_ws.abs().amax(dim=(0, 1, 3)).min()) > 0          # every one of the num_ws slots receives a gradient
    np.savez_compressed(os.path.join(OUT, "wplus_tiny.npz"), z=z.numpy(), ws=ws.numpy(), img_ws=img_ws.numpy(),
                        loss_ws=np.float32(loss_ws.item()), grad_ws=grad_ws.numpy(),
                        att_shape=np.array(att_ws.shape), att_sub=att_ws[:, :, :, 0, 3::8, 5::8].numpy(),
                        img_cut=img_cut.numpy(), ws_cut=ws_cut.numpy(), img_psi=img_psi.numpy(), ws_psi=ws_psi.numpy(),
                        img_synthesis_subnet=G(ws=ws, noise_mode="const", subnet="synthesis").numpy())
    print("wplus_tiny", tuple(att_ws.shape))


def grad_full_target(res=1024):
    """Deterministic smooth target of the full-size gradient fixture (formula shared with tests/test_hip_gradient.py)."""
    y, x = np.meshgrid(np.arange(res, dtype=np.float64), np.arange(res, dtype=np.float64), indexing="ij")
    return np.stack([0.5 * np.sin(2 * np.pi * (x * (c + 1) + y) / res) for c in range(3)])[None].astype(np.float32)


def gold_grad_full(ref):
    """Gradient-mode fixture at full size: d MSE(G(z), target) / dz through the REFERENCE module's autograd at 1024^2."""
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    sd = make_state_dict(FULL1024, seed=0)
    G = build_reference_generator(ref, FULL1024, sd)
    G.synthesis.list2tensor = lambda att_list, device: torch.zeros([1])
    torch.set_num_threads(8)
    z = torch.from_numpy(synthetic_latents(FULL1024, 1, seed=1000)).requires_grad_(True)
    target = torch.from_numpy(grad_full_target(1024))
    loss = (G(z, None, noise_mode="const")[0] - target).square().mean()
    (gz,) = torch.autograd.grad(loss, z)
    np.savez_compressed(os.path.join(OUT, "grad_full1024.npz"), z=z.detach().numpy(), loss=np.float64(loss.item()), grad_z=gz.numpy())
    print("grad_full1024: loss", loss.item(), "|grad|max", float(gz.abs().max()))


def gold_config0_256(ref):
    """BASELINE configs[0]: one 256x256 face, 50-step MSE-only literal projection (the reference's CPU-runnable case).
    The target is a uint8 image pushed through ToTensor+Normalize like image_transform (...sqz_MSE.py:89-108)."""
    from morphganformer_amd.synth_weights import SMALL256, make_state_dict, synthetic_latents
    cfg = SMALL256
    sd = make_state_dict(cfg, seed=0)
    G = build_reference_generator(ref, cfg, sd)
    G.synthesis.list2tensor = lambda att_list, device: torch.zeros([1])
    steps = 50
    rng = np.random.Generator(np.random.PCG64(2560))
    samples = torch.from_numpy(rng.standard_normal((10000, cfg.k, cfg.z_dim)).astype(np.float32))
    latent_mean = samples.mean(0)
    latent_std = ((samples - latent_mean).pow(2).sum() / samples.shape[0]) ** 0.5
    timg = G(torch.from_numpy(synthetic_latents(cfg, 1, seed=1002)), None, noise_mode="const")[0].clamp(-1, 1)
    target_u8 = ((timg[0] + 1) * 127.5).round().clamp(0, 255).to(torch.uint8)
    target = target_u8.float().div(255).sub(0.5).div(0.5)[None]
    eps = rng.standard_normal((steps, 1, cfg.k, cfg.z_dim)).astype(np.float32)
    mse = torch.nn.MSELoss()
    latent_in = latent_mean[None].clone()
    min_loss, best, best_step = 100.0, None, -1
    losses = np.zeros(steps, np.float64)
    sigmas, latents_n = [], []
    for i in range(steps):
        t = i / steps
        sigma = latent_std * 0.05 * max(0, 1 - t / 0.75) ** 2
        latent_n = latent_in + torch.from_numpy(eps[i]) * sigma.item()
        sigmas.append(sigma.item()); latents_n.append(latent_n.numpy().copy())
        img = G(latent_n, 0.7, noise_mode="const")[0]
        total = 1.0 * mse(img, target)
        losses[i] = float(total)
        if float(total) < min_loss:
            min_loss, best, best_step = float(total), latent_n.clone(), i
    np.savez_compressed(os.path.join(OUT, "loop_config0_256.npz"), latent_mean=latent_mean.numpy(),
                        latent_std=np.float32(latent_std.item()), target_u8=target_u8.numpy(), eps=eps, losses=losses,
                        best_latent=best.numpy(), best_step=np.int64(best_step), best_loss=np.float64(min_loss),
                        sigmas=np.array(sigmas, np.float64), latents_n=np.stack(latents_n))


def gold_morph_tiny(ref, G):
    """BASELINE config 4's rendering half: 11 linear morphs of two latents through G(dw, psi) (1024_merge_morph_2.py:83-86;
    psi is passed positionally, i.e. into `c`, so no truncation is applied)."""
    from morphganformer_amd.synth_weights import TINY, synthetic_latents
    w1 = synthetic_latents(TINY, 1, seed=2001)
    w2 = synthetic_latents(TINY, 1, seed=2002)
    alphas = np.linspace(0.0, 1.0, 11)
    imgs, lats = [], []
    for a in alphas:
        dw = 0.5 * w1 + 0.5 * w2 if a == 0.5 else np.float32(1.0 - a) * w1 + np.float32(a) * w2
        imgs.append(G(torch.from_numpy(dw), 0.7, noise_mode="const")[0][0].numpy())
        lats.append(dw)
    # the keyword form used by 1024_generate.py:35 does apply the truncation
    img_psi = G(torch.from_numpy(w1), truncation_psi=0.7, noise_mode="const")[0].numpy()
    np.savez_compressed(os.path.join(OUT, "morph_tiny.npz"), w1=w1, w2=w2, alphas=alphas, latents=np.stack(lats),
                        images=np.stack(imgs), img_w1_psi07=img_psi)


def gold_iresnet():
    """The vendored face embedder (backbones/iresnet.py, pure torch): iresnet18 with the build's seeded state on a seeded
    112x112 batch -> embedding and per-stage statistics."""
    sys.path.insert(0, REF)
    from backbones import iresnet as ref_iresnet
    from morphganformer_amd.iresnet import random_state
    sd = random_state(18, seed=0)
    net = ref_iresnet.iresnet18().eval()
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    rng = np.random.Generator(np.random.PCG64(112))
    x = rng.uniform(-1, 1, (2, 3, 112, 112)).astype(np.float32)
    stats = {}
    hooks = [getattr(net, f"layer{i}").register_forward_hook(
        lambda m, inp, out, i=i: stats.__setitem__(i, (float(out.double().mean()), float(out.double().square().mean().sqrt()))))
        for i in range(1, 5)]
    with torch.no_grad():
        emb = net(torch.from_numpy(x))
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(OUT, "iresnet18.npz"), x=x, embedding=emb.numpy(),
                        layer_mean=np.array([stats[i][0] for i in range(1, 5)]), layer_rms=np.array([stats[i][1] for i in range(1, 5)]))


def gold_loss_kats(ref):
    out = {}
    wing = ref.wing_loss.WingLoss()
    awing = ref.adaptive_wing_loss.AdaptiveWingLoss()
    out["wing_ones_zeros"] = np.float64(wing(torch.zeros(2, 68, 64, 64), torch.ones(2, 68, 64, 64)).item())
    p = torch.tensor([[1.0, 12.0], [3.0, 0.0]], dtype=torch.float64)
    t = torch.tensor([[0.0, 0.0], [3.0, 20.0]], dtype=torch.float64)
    out["wing_small_pred"], out["wing_small_target"] = p.numpy(), t.numpy()
    out["wing_small"] = np.float64(wing(p, t).item())
    out["awing_ones_zeros"] = np.float64(awing(torch.zeros(68, 2), torch.ones(68, 2)).item())
    rng = np.random.Generator(np.random.PCG64(21))
    a = rng.integers(0, 1024, size=(68, 2)).astype(np.float64)
    b = a + rng.integers(-30, 31, size=(68, 2))
    out["wing_rand_pred"], out["wing_rand_target"] = a, b
    out["wing_rand"] = np.float64(wing(torch.from_numpy(a), torch.from_numpy(b)).item())
    np.savez_compressed(os.path.join(OUT, "loss_kats.npz"), **out)


def gold_lin_heads():
    """The vendored LPIPS 1x1 'lin' heads are data files of the reference (lpips/weights/v0.1/*.pth)."""
    for net in ("squeeze", "alex", "vgg"):
        sd = torch.load(os.path.join(REF, "lpips", "weights", "v0.1", net + ".pth"), map_location="cpu")
        arrs = {k.replace(".model.1.weight", ""): v.reshape(-1).numpy() for k, v in sd.items()}
        np.savez_compressed(os.path.join(OUT, f"lpips_lin_{net}.npz"), **arrs)


def import_reference_lpips():
    """Import the reference's `lpips` package.  Its modules import three absent third-party packages at module scope
    (skimage: lpips/__init__.py:7, networks_basic.py:11, dist_model.py:16; IPython: networks_basic.py:12 ...; torchvision:
    pretrained_networks.py:3) -> IMPORT-ONLY stub modules, exactly like termcolor/seaborn above: they define the imported names and
    nothing else, and none of them is ever called (the torchvision backbones are never constructed here)."""
    sys.path.insert(0, REF)

    def stub(name, **attrs):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
        return sys.modules[name]

    def absent(*a, **k):
        raise RuntimeError("import-only stub: this third-party function is not available offline")

    sk = stub("skimage")
    sk.measure = stub("skimage.measure", compare_ssim=absent)
    sk.color = stub("skimage.color")
    sk.transform = stub("skimage.transform")
    stub("IPython", embed=absent)
    tvm = stub("torchvision")
    tvm.models = stub("torchvision.models")
    import lpips
    from lpips import networks_basic
    return lpips, networks_basic


class _TapSource(torch.nn.Module):
    """Stands where PNetLin keeps its torchvision backbone (`self.net`, networks_basic.py:52): returns pre-computed tap tensors, keyed by
    the identity of the (scaled) input, so that the reference's PNetLin.forward runs verbatim on injected feature maps."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x)


def _reference_pnetlin(nb, net, feat_fn):
    """A reference PNetLin (networks_basic.py:26-92) with the vendored lin heads loaded the way DistModel.initialize does
    (dist_model.py:63-75: use_dropout=True, load_state_dict(strict=False), eval) and `feat_fn` in place of the torchvision
    backbone.  PNetLin.__init__ would construct tv.<net>(pretrained=...) -- absent offline -- so the attributes its forward reads
    are set here to the values __init__ gives them (:29-62); forward, ScalingLayer, NetLinLayer are the reference's own code."""
    chns = {"squeeze": [64, 128, 256, 384, 384, 512, 512], "vgg": [64, 128, 256, 512, 512], "alex": [64, 192, 384, 256, 256]}[net]
    m = nb.PNetLin.__new__(nb.PNetLin)
    torch.nn.Module.__init__(m)
    m.pnet_type, m.pnet_tune, m.pnet_rand, m.spatial, m.lpips, m.version = net, False, False, False, True, "0.1"
    m.scaling_layer = nb.ScalingLayer()
    m.chns, m.L = chns, len(chns)
    m.net = _TapSource(feat_fn)
    m.lins = []
    for i, c in enumerate(chns):
        lin = nb.NetLinLayer(c, use_dropout=True)
        setattr(m, f"lin{i}", lin)
        m.lins.append(lin)
    state = torch.load(os.path.join(REF, "lpips", "weights", "v0.1", net + ".pth"), map_location="cpu")
    missing, unexpected = m.load_state_dict(state, strict=False)
    assert not unexpected and not [k for k in missing if k.startswith("lin")], (missing, unexpected)
    return m.eval()


def gold_lpips_dist():
    """LPIPS distance half run through the REFERENCE's own code (lpips/networks_basic.py:64-111, lpips/__init__.py:44-46):
      (1) per net in {squeeze, vgg, alex}: seeded tap tensors (one pair per tap, the net's channel counts) -> reference
          normalize_tensor, squared difference, NetLinLayer (vendored weights, eval => dropout is identity), spatial_average, sum:
          `PNetLin.forward(retPerLayer=True)` with the tap tensors standing where the backbone output would be;
      (2) ScalingLayer on a seeded image;
      (3) squeeze: the whole PNetLin.forward on two seeded 64x64 images with the ORACLE's restated SqueezeNet1.1 topology + seeded
          random weights as the injected backbone (the torchvision backbone itself stays unpinned)."""
    lp, nb = import_reference_lpips()
    from oracle.loss_ref import squeeze_backbone_random, squeeze_features_ref
    rng = np.random.Generator(np.random.PCG64(314))
    out = {}
    sizes = {"squeeze": [(9, 9), (5, 5), (3, 3), (2, 2), (2, 2), (2, 2), (2, 2)], "vgg": [(8, 8), (4, 4), (4, 4), (2, 2), (2, 2)],
             "alex": [(7, 7), (3, 3), (3, 3), (3, 3), (3, 3)]}
    with torch.no_grad():
        for net, hw in sizes.items():
            chns = {"squeeze": [64, 128, 256, 384, 384, 512, 512], "vgg": [64, 128, 256, 512, 512], "alex": [64, 192, 384, 256, 256]}[net]
            # post-ReLU-like taps: non-negative, some all-zero pixels (exercises the eps of normalize_tensor), n = 2 samples
            t0 = [np.maximum(rng.standard_normal((2, c, h, w)), 0).astype(np.float32) for c, (h, w) in zip(chns, hw)]
            t1 = [np.maximum(rng.standard_normal((2, c, h, w)), 0).astype(np.float32) for c, (h, w) in zip(chns, hw)]
            for t in t0 + t1:
                t[:, :, 0, 0] = 0
            seq = iter([t0, t1])
            m = _reference_pnetlin(nb, net, lambda x: [torch.from_numpy(a) for a in next(seq)])
            img = torch.zeros(2, 3, 8, 8)
            val, res = m.forward(img, img, retPerLayer=True)
            for i in range(len(chns)):
                out[f"{net}_tap0_{i}"], out[f"{net}_tap1_{i}"] = t0[i], t1[i]
                out[f"{net}_res_{i}"] = res[i].numpy()
            out[f"{net}_unit0_1"] = lp.normalize_tensor(torch.from_numpy(t0[1])).numpy()
            out[f"{net}_val"] = val.numpy()
        x = rng.uniform(-1, 1, (2, 3, 5, 7)).astype(np.float32)
        out["scale_in"], out["scale_out"] = x, nb.ScalingLayer()(torch.from_numpy(x)).numpy()
        # (3) full forward, oracle topology injected
        bb = squeeze_backbone_random(0)
        m = _reference_pnetlin(nb, "squeeze", lambda xs: squeeze_features_ref(bb, xs))
        a = rng.uniform(-1, 1, (2, 3, 64, 64)).astype(np.float32)
        b = np.clip(a + 0.3 * rng.standard_normal(a.shape), -1, 1).astype(np.float32)
        val, res = m.forward(torch.from_numpy(a), torch.from_numpy(b), retPerLayer=True)
        out["full_in0"], out["full_in1"], out["full_val"] = a, b, val.numpy()
        out["full_res"] = np.stack([r.numpy().reshape(-1) for r in res])
        # the module-level entry the drivers call: PerceptualLoss.forward -> DistModel.forward -> PNetLin.forward (lpips/__init__.py:26-41,
        # dist_model.py:110-118), with normalize=True mapping [0,1] images to [-1,1]
        pl = lp.PerceptualLoss.__new__(lp.PerceptualLoss)
        torch.nn.Module.__init__(pl)
        dm = lp.dist_model.DistModel()
        dm.net = m
        pl.model, pl.use_gpu, pl.spatial, pl.gpu_ids = dm, False, False, [0]
        out["full_val_normalize"] = pl.forward(torch.from_numpy((a + 1) / 2), torch.from_numpy((b + 1) / 2), normalize=True).numpy()
    np.savez_compressed(os.path.join(OUT, "lpips_dist.npz"), **out)
    print("lpips_dist: full val", out["full_val"].reshape(-1))


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    if only == ["lpips"]:
        gold_lpips_dist()
        return
    ref = import_reference()
    torch.manual_seed(0)
    if only:
        if "lpips" in only:
            gold_lpips_dist()
        # regenerate a subset without touching the other fixtures
        if "config0" in only:
            gold_config0_256(ref)
        if "loop" in only:
            from morphganformer_amd.synth_weights import TINY, make_state_dict
            gold_loop_tiny(ref, build_reference_generator(ref, TINY, make_state_dict(TINY, seed=0)))
        if "att" in only:
            gold_att_tiny(ref)
        if "copies" in only:
            from morphganformer_amd.synth_weights import TINY, make_state_dict
            gold_loop_copies_tiny(ref, build_reference_generator(ref, TINY, make_state_dict(TINY, seed=0)))
        if "wplus" in only:
            gold_wplus_tiny(ref)
        if "gradfull" in only:
            gold_grad_full(ref)
        if "full" in only:
            gold_generator_full(ref)
        if "iresnet" in only:
            gold_iresnet()
        if "morph" in only:
            from morphganformer_amd.synth_weights import TINY, make_state_dict
            gold_morph_tiny(ref, build_reference_generator(ref, TINY, make_state_dict(TINY, seed=0)))
        return
    gold_bias_act(ref)
    gold_upfirdn2d(ref)
    gold_modconv(ref)
    gold_loss_kats(ref)
    gold_lin_heads()
    G, _ = gold_generator_tiny(ref)
    gold_loop_tiny(ref, G)
    gold_loop_copies_tiny(ref, G)
    gold_morph_tiny(ref, G)
    gold_att_tiny(ref)
    gold_wplus_tiny(ref)
    gold_config0_256(ref)
    gold_iresnet()
    gold_lpips_dist()
    if "--no-full" not in sys.argv:
        gold_generator_full(ref)
        gold_grad_full(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
