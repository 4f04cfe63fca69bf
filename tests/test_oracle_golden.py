"""CPU: the oracle (oracle/*.py) is pinned against outputs of the REFERENCE itself (tests/golden/*.npz, produced by
oracle/make_golden.py from /root/reference in the build container) and against the reference's own known answers
(SURVEY.md section 8c).  Everything here is float32 torch-CPU; the restatement reproduces the reference bit for bit on
this container's torch build, so the tolerances below are only slack for other BLAS/oneDNN builds."""
import math

import numpy as np
import pytest
import torch

from morphganformer_amd.synth_weights import FULL1024, TINY, make_state_dict
from oracle import loss_ref
from oracle.generator_ref import generator_ref, mapping_ref, to_torch_state
from oracle.ops_ref import (ACT_NAMES, bias_act_grad_ref, bias_act_ref, conv2d_resample_ref, modulated_conv2d_ref,
                            setup_filter_ref, upfirdn2d_ref)

TOL = 2e-6


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("clamp", [None, 0.5])
def test_bias_act_ref_vs_reference(golden, clamp):
    g = golden("ops_bias_act.npz")
    x, b, dy, ddx = (torch.from_numpy(g[k]) for k in ("x", "b", "dy", "ddx"))
    for act in ACT_NAMES:
        tag = f"{act}_c{'none' if clamp is None else clamp}"
        y = bias_act_ref(x, b, dim=1, act=act, clamp=clamp)
        assert rel(y, g[f"y_{tag}"]) < TOL, act
        dx = bias_act_grad_ref(dy, x, b, y, dim=1, act=act, clamp=clamp, order=1)
        assert rel(dx, g[f"dx_{tag}"]) < 5e-6, act
        d2 = bias_act_grad_ref(dy, x, b, y, dim=1, act=act, clamp=clamp, order=2, ddx=ddx)
        if np.abs(g[f"d2_{tag}"]).max() > 0:
            assert rel(d2, g[f"d2_{tag}"]) < 2e-5, act
        else:
            assert float(d2.abs().max()) == 0.0
    assert rel(bias_act_ref(torch.from_numpy(g["x2"]), torch.from_numpy(g["b2"]), dim=0, act="lrelu", alpha=0.3, gain=1.7), g["y2"]) < TOL


def test_bias_act_ref_relu_gradient_at_exactly_zero():
    """bias_act.py:17 is `torch.nn.functional.relu`: autograd gives it (and the plugin's `y > 0` test, bias_act.cu) the gradient 0 AT zero -- `clamp_min`,
    which the restatement used until a round-6 soak run drew x + b == 0 in float16, gives 1."""
    x = torch.tensor([0.0, -1.0, 2.0, 0.0], dtype=torch.float64, requires_grad=True)
    b = torch.tensor([0.0, 0.0, 0.0, 0.0], dtype=torch.float64)
    (g,) = torch.autograd.grad(bias_act_ref(x.reshape(1, 4), b, dim=1, act="relu").sum(), x)
    assert g.tolist() == [0.0, 0.0, math.sqrt(2), 0.0]


def test_upfirdn2d_ref_vs_reference(golden):
    from oracle.make_golden import UPFIRDN_CASES
    g = golden("ops_upfirdn2d.npz")
    for name, shape, taps, up, down, pad, gain, flip in UPFIRDN_CASES:
        if taps is not None:
            assert torch.equal(setup_filter_ref(taps), torch.from_numpy(g[f"f_{name}"])), name
        y = upfirdn2d_ref(torch.from_numpy(g[f"x_{name}"]), torch.from_numpy(g[f"f_{name}"]), up=up, down=down, padding=pad,
                          flip_filter=flip, gain=gain)
        assert tuple(y.shape) == g[f"y_{name}"].shape and rel(y, g[f"y_{name}"]) < TOL, name


def test_modconv_ref_vs_reference(golden):
    g = golden("ops_modconv.npz")
    x, w, s, f = (torch.from_numpy(g[k]) for k in ("x", "w", "s", "f"))
    for up in (1, 2):
        for demod in (True, False):
            y = modulated_conv2d_ref(x, w, s, up=up, padding=1, resample_kernel=f, demodulate=demod, flip_weight=(up == 1))
            assert rel(y, g[f"y_up{up}_demod{int(demod)}"]) < TOL
    y = conv2d_resample_ref(x, torch.from_numpy(g["w1"]), f=f, up=2, padding=0, flip_weight=False)
    assert rel(y, g["y_skip_up2"]) < TOL


def test_generator_ref_vs_reference_tiny(golden):
    g = golden("gen_tiny.npz")
    sd = to_torch_state(make_state_dict(TINY, 0))
    z = torch.from_numpy(g["z"])
    taps = {}
    img = generator_ref(sd, z, TINY, "const", taps=taps)
    assert rel(taps["ws"], g["ws"]) < TOL
    for r in TINY.block_resolutions:
        assert rel(taps[f"synthesis.b{r}"], g[f"tap_b{r}"]) < 1e-5, r
    assert rel(img, g["img_const"]) < 1e-5
    assert rel(generator_ref(sd, z, TINY, "none"), g["img_none"]) < 1e-5
    noises = {k[len("noise_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("noise_")}
    assert rel(generator_ref(sd, z, TINY, "inject", noises), g["img_inject"]) < 1e-5
    for key in ("b4.conv1", "b16.conv0", "b64.conv1"):
        p = taps[f"synthesis.{key}:probs"].numpy()
        assert rel(p, g["probs_" + key]) < 1e-5
        assert np.array_equal(p.argmax(-1), g["probs_" + key].argmax(-1))
    # gradient-mode oracle: d mean(img^2) / dz through the restatement == through the reference module
    zg = z.clone().requires_grad_(True)
    loss = generator_ref(sd, zg, TINY, "const").square().mean()
    (gz,) = torch.autograd.grad(loss, zg)
    assert abs(float(loss.detach()) - float(g["loss_sq"])) < 1e-5 * float(g["loss_sq"])
    assert rel(gz, g["grad_z"]) < 1e-4


def test_mapping_ref_properties():
    sd = to_torch_state(make_state_dict(TINY, 0))
    torch.manual_seed(0)
    z = torch.randn(3, TINY.k, TINY.z_dim)
    w = mapping_ref(sd, z, TINY)
    assert tuple(w.shape) == (3, TINY.k, TINY.w_dim)
    # samples are independent: evaluating one alone gives the same row
    wmax = float(w.abs().max())                       # (~ 18: the gates are relative to it -- float32 rounding of the restatement itself is 1e-6 of that)
    assert float((mapping_ref(sd, z[1:2], TINY) - w[1:2]).abs().max()) < 5e-7 * wmax
    # the joint normalisation makes the local path invariant to a positive rescale of the local components
    z2 = z.clone(); z2[:, :-1] *= 3.0
    assert float((mapping_ref(sd, z2, TINY)[:, :-1] - w[:, :-1]).abs().max()) < 5e-6 * wmax


def test_loss_kats(golden):
    g = golden("loss_kats.npz")
    assert abs(float(loss_ref.wing_loss_ref(torch.zeros(2, 68, 64, 64), torch.ones(2, 68, 64, 64))) - float(g["wing_ones_zeros"])) < 1e-6
    assert abs(float(g["wing_ones_zeros"]) - 4.054649829864502) < 1e-6            # SURVEY.md 8c [probe]
    v = loss_ref.wing_loss_ref(torch.from_numpy(g["wing_small_pred"]), torch.from_numpy(g["wing_small_target"]))
    assert abs(float(v) - float(g["wing_small"])) < 1e-12 and abs(float(v) - 12.972460116410685) < 1e-9
    v = loss_ref.wing_loss_ref(torch.from_numpy(g["wing_rand_pred"]), torch.from_numpy(g["wing_rand_target"]))
    assert abs(float(v) - float(g["wing_rand"])) < 1e-10
    v = loss_ref.adaptive_wing_loss_ref(torch.zeros(68, 2), torch.ones(68, 2))
    assert abs(float(v) - float(g["awing_ones_zeros"])) < 1e-5 and abs(float(v) - 10.259384155273438) < 1e-4


def test_schedule_kats():
    lr = [loss_ref.get_lr_ref(t, 0.01) for t in (0, .025, .05, .5, .875, .999)]
    for got, want in zip(lr, (0, .005, .01, .01, .005, 3.9e-07)):
        assert abs(got - want) < 1e-4 * max(want, 1e-3) + 2e-8
    ns = [loss_ref.noise_strength_ref(t, 1.0, 1.0, 0.75) for t in (0, .25, .5, .75)]
    assert np.allclose(ns, (1, 4 / 9, 1 / 9, 0))
    torch.manual_seed(0)
    mean, std = loss_ref.latent_stats_ref(torch.randn(10000, 17, 32))
    assert abs(float(std) - 23.316194534) < 2e-4                                  # SURVEY.md 8c [probe], seed 0 CPU
    img = np.array([[-1.0, 1.0, 0.0, 0.0039, 2.0, -3.0]], np.float32)          # 2-D (HW) path of to_pil
    assert loss_ref.to_uint8_ref(img)[0].tolist() == [0, 255, 128, 128, 255, 0]    # rint half-to-even at 127.5


def test_projection_literal_ref_vs_reference_run(golden):
    g = golden("loop_tiny.npz")
    sd = to_torch_state(make_state_dict(TINY, 0))
    target = torch.from_numpy(g["target"])
    lm_t, lm_s = g["lm_target"], g["lm_steps"]
    steps = int(g["eps"].shape[0])

    def loss_fn(i, img):
        return float(0.01 * loss_ref.wing_loss_ref(torch.from_numpy(lm_s[i]), torch.from_numpy(lm_t)) + loss_ref.mse_ref(img, target))

    best, bstep, bloss, losses = loss_ref.projection_literal_ref(
        lambda z: generator_ref(sd, z, TINY, "const"), loss_fn, torch.from_numpy(g["latent_mean"]), float(g["latent_std"]),
        torch.from_numpy(g["eps"]), steps)
    assert bstep == int(g["best_step"])
    assert np.array_equal(best.numpy(), g["best_latent"])
    assert np.allclose(losses, g["losses"], rtol=1e-5)        # float32 MSE reduction order (nn.MSELoss vs square().mean())


def test_lpips_ref_properties():
    bb = loss_ref.squeeze_backbone_random(0)
    lin = np.load("morphganformer_amd/weights/lpips_lin_squeeze.npz")
    lins = [torch.from_numpy(lin[f"lin{i}"]) for i in range(7)]
    assert all(float(l.min()) >= 0 for l in lins) and [l.numel() for l in lins] == loss_ref.SQUEEZE_CHNS
    torch.manual_seed(1)
    a = torch.rand(1, 3, 67, 67) * 2 - 1
    b = torch.rand(1, 3, 67, 67) * 2 - 1
    taps = loss_ref.squeeze_features_ref(bb, a)
    assert [t.shape[1] for t in taps] == loss_ref.SQUEEZE_CHNS
    assert [t.shape[2] for t in taps] == [33, 16, 8, 4, 4, 4, 4]                   # stride-2 conv then three ceil-mode pools
    d_ab, d_ba, d_aa = (float(loss_ref.lpips_ref(bb, lins, x, y)) for x, y in ((a, b), (b, a), (a, a)))
    assert d_aa == 0.0 and d_ab > 0 and abs(d_ab - d_ba) < 1e-6 * d_ab             # identity and symmetry


@pytest.mark.slow
def test_generator_ref_full_vs_reference(golden):
    """Full 1024^2 forward of the restatement (about 1.5 s on 8 cores, 1 GB peak) vs reference samples/checksums."""
    g = golden("gen_full1024.npz")
    sd = to_torch_state(make_state_dict(FULL1024, 0))
    taps = {}
    with torch.no_grad():
        img = generator_ref(sd, torch.from_numpy(g["z"]), FULL1024, "const", taps=taps)
    assert rel(img.reshape(-1)[torch.from_numpy(g["idx"])], g["pixels"]) < 1e-5
    assert abs(float(img.double().mean()) - float(g["img_mean"])) < 1e-6
    for res, m, r in zip(g["block_res"], g["block_mean"], g["block_rms"]):
        t = taps[f"synthesis.b{int(res)}"].double()
        assert abs(float(t.mean()) - m) < 1e-5 * r and abs(float(t.square().mean().sqrt()) - r) < 1e-5 * r


def _att_decided(g, key, slot):
    F = g["argmax_" + key].shape[1]
    return np.unpackbits(g["decided_" + key][slot])[:F].astype(bool)


@pytest.mark.slow
def test_attention_assignments_full_vs_reference(golden):
    """SURVEY 8d's integer gate at FULL size for the restatement: the per-pixel argmax latent assignment of all 11 attention layers of
    the 1024^2 model (networks.py:505-524,776-792) against the reference module's own (att_full1024.npz: forward hooks on its
    TransformerLayers), exact wherever the reference's top-2 margin exceeds 1e-4, and 256 sampled probability rows per layer."""
    g = golden("att_full1024.npz")
    sd = to_torch_state(make_state_dict(FULL1024, 0))
    assert len(g["layers"]) == 11
    for slot in range(2):
        taps = {}
        with torch.no_grad():
            generator_ref(sd, torch.from_numpy(g["z"][slot:slot + 1]), FULL1024, "const", taps=taps)
        for key in g["layers"]:
            p = taps[f"synthesis.{key}:probs"].reshape(-1, FULL1024.k - 1).numpy()
            dec = _att_decided(g, key, slot)
            assert dec.mean() > 0.99, key
            assert np.array_equal(p.argmax(-1)[dec], g["argmax_" + key][slot][dec]), key
            assert np.abs(p[g["rows_" + key]] - g["probs_" + key][slot]).max() < 1e-5, key


def test_morph_ref_vs_reference(golden):
    """BASELINE config 4's rendering half: the oracle generator on the 11 blended latents the reference rendered."""
    g = golden("morph_tiny.npz")
    sd = to_torch_state(make_state_dict(TINY, 0))
    for i in (0, 5, 10):
        img = generator_ref(sd, torch.from_numpy(g["latents"][i]), TINY, "const")
        assert np.allclose(img[0].numpy(), g["images"][i], atol=2e-5, rtol=0)
    assert np.array_equal(g["latents"][5], 0.5 * g["w1"] + 0.5 * g["w2"])


@pytest.mark.slow
def test_projection_literal_ref_config0_256(golden):
    """BASELINE configs[0] (256^2, 50 steps, MSE only) -- the oracle loop against the reference run."""
    from morphganformer_amd.synth_weights import SMALL256
    g = golden("loop_config0_256.npz")
    sd = to_torch_state(make_state_dict(SMALL256, 0))
    target = torch.from_numpy(g["target_u8"]).float().div(255).sub(0.5).div(0.5)[None]
    steps = 10          # the first 10 of the 50 recorded steps keep the CPU suite short
    best, bstep, bloss, losses = loss_ref.projection_literal_ref(
        lambda z: generator_ref(sd, z, SMALL256, "const"), lambda i, img: float(loss_ref.mse_ref(img, target)),
        torch.from_numpy(g["latent_mean"]), float(g["latent_std"]), torch.from_numpy(g["eps"]), steps, total_steps=50)
    assert np.allclose(losses[:steps], g["losses"][:steps], rtol=1e-5)


def test_iresnet_ref_vs_reference_module(golden):
    """oracle/embed_ref.py against backbones/iresnet.py (iresnet18, seeded state, eval mode): bit for bit on this torch build."""
    from morphganformer_amd.iresnet import block_table, random_state
    from oracle.embed_ref import iresnet_ref
    g = golden("iresnet18.npz")
    sd = {k: torch.from_numpy(v) for k, v in random_state(18, 0).items()}
    taps = {}
    with torch.no_grad():
        e = iresnet_ref(sd, torch.from_numpy(g["x"]), 18, taps)
    assert np.array_equal(e.numpy(), g["embedding"])
    rows = block_table(18)
    assert len(rows) == 8 and [r[3] for r in rows] == [2, 1, 2, 1, 2, 1, 2, 1] and rows[-1][0] == "layer4.1"
    for i, last in enumerate(("layer1.1", "layer2.1", "layer3.1", "layer4.1")):
        assert abs(float(taps[last].double().square().mean().sqrt()) - g["layer_rms"][i]) < 1e-9 * g["layer_rms"][i]
    assert sum(len(v) for v in ([1] * 3,)) == 3 and len(block_table(50)) == 24 and len(block_table(100)) == 49


def test_generator_ref_gradient_full_vs_reference(golden):
    """The restatement's autograd at full size against the reference module's (tests/golden/grad_full1024.npz)."""
    from oracle.make_golden import grad_full_target
    g = golden("grad_full1024.npz")
    sd = to_torch_state(make_state_dict(FULL1024, 0))
    z = torch.from_numpy(g["z"]).requires_grad_(True)
    target = torch.from_numpy(grad_full_target(1024))
    loss = (generator_ref(sd, z, FULL1024, "const") - target).square().mean()
    (gz,) = torch.autograd.grad(loss, z)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * float(g["loss"])
    assert rel(gz, g["grad_z"]) < 1e-4


def test_psnr_script_pipeline_known_answer_by_hand():
    """1024_example_PSNR.py:150-158 transcribed (oracle.loss_ref.psnr_script_ref): the candidate reaches `psnr` in C-H-W order, the target in
    H-W-C order.  x = arange(12) as [1,3,2,2] against ITSELF: the aligned PSNR is infinite (MSE 0); the script pairs the streams
    [0..11] and [0,4,8,1,5,9,2,6,10,3,7,11]: squared differences 0+9+36+4+1+16+16+1+4+36+9+0 = 132, mean 11 -> 10 log10(255^2 / 11)."""
    x = torch.arange(12, dtype=torch.float32).reshape(1, 3, 2, 2)
    got = float(loss_ref.psnr_script_ref(x, x))
    assert abs(got - 10 * math.log10(65025.0 / 11.0)) < 1e-5, got
    with np.errstate(divide="ignore"):
        assert np.isinf(loss_ref.psnr_ref(x.numpy(), x.numpy()))
    # the same pairing expressed as one permuted copy of the target (what the engine does): aligned PSNR against the H-W-C stream re-read as C-H-W
    t = torch.randn(1, 3, 5, 4)
    y = torch.randn(1, 3, 5, 4)
    t_script = t.permute(0, 2, 3, 1).contiguous().view(t.shape)
    assert abs(float(loss_ref.psnr_script_ref(y, t)) - float(loss_ref.psnr_ref(y.numpy(), t_script.numpy()))) < 1e-5


def test_latent_copies_loop_vs_reference_run(golden):
    """projection_example_v2_percept.py:131-203 (18 noisy copies of the latent averaged by torch.mean, min_loss from 1.0): the restated loop
    against the run of the script's own tensor statements on the reference generator (loop_copies_tiny.npz) -- every step's averaged latent
    bit for bit, every loss, the selection; and the numpy restatement of torch's summation order against torch.mean itself for 1..40 and
    255 rows (the order the device kernel follows)."""
    g = golden("loop_copies_tiny.npz")
    sd = to_torch_state(make_state_dict(TINY, 0))
    steps, copies = g["eps"].shape[0], int(g["copies"])
    target = torch.from_numpy(g["target"])
    seen = []

    def gen(z):
        seen.append(z.numpy().copy())
        with torch.no_grad():
            return generator_ref(sd, z, TINY, "const")
    lat, bstep, bloss, losses = loss_ref.projection_literal_ref(
        gen, lambda i, img: float(0.01 * loss_ref.mse_ref(img, target)), torch.from_numpy(g["latent_mean"]).reshape(TINY.k, TINY.z_dim),
        float(g["latent_std"]), torch.from_numpy(g["eps"]), steps, min_loss_init=1.0, copies=copies)
    assert np.array_equal(np.stack(seen), g["im_latents"])
    assert bstep == int(g["best_step"]) and np.array_equal(lat.numpy(), g["best_latent"])
    assert rel(np.array(losses), g["losses"]) < 1e-5
    rng = np.random.Generator(np.random.PCG64(7))
    for n in list(range(1, 41)) + [255]:
        x = (rng.standard_normal((n, 96)) * 3 + 0.7).astype(np.float32)      # (a multiple of the vector width: see mean_rows_torch_order_ref)
        assert np.array_equal(loss_ref.mean_rows_torch_order_ref(x), torch.mean(torch.from_numpy(x)[None], 1)[0].numpy()), n


def test_winograd_f2x2_3x3_identity():
    """The transform matrices csrc/wino.hip hard-codes (Lavin & Gray F(2x2,3x3)): A^T [(G g G^T) . (B^T d B)] A equals the 3x3
    correlation of a 4x4 patch, in float64, for random data -- pins the algebra the HIP kernels implement."""
    rng = np.random.Generator(np.random.PCG64(1))
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    for _ in range(5):
        d, g = rng.standard_normal((4, 4)), rng.standard_normal((3, 3))
        y = At @ ((G @ g @ G.T) * (Bt @ d @ Bt.T)) @ At.T
        ref = np.array([[(g * d[i:i + 3, j:j + 3]).sum() for j in range(2)] for i in range(2)])
        assert np.abs(y - ref).max() < 1e-12


LPIPS_CHNS = {"squeeze": [64, 128, 256, 384, 384, 512, 512], "vgg": [64, 128, 256, 512, 512], "alex": [64, 192, 384, 256, 256]}


def test_lpips_distance_half_vs_reference(golden):
    """tests/golden/lpips_dist.npz holds outputs of the reference's own lpips code (networks_basic.py:64-111 PNetLin.forward,
    ScalingLayer, NetLinLayer with the vendored weights, lpips/__init__.py:26-46) on injected tap tensors: the oracle's distance
    half, its ScalingLayer and the whole lpips_ref (restated SqueezeNet topology, same seeded weights) must reproduce them."""
    g = golden("lpips_dist.npz")
    for net, chns in LPIPS_CHNS.items():
        lin = golden(f"lpips_lin_{net}.npz")
        lins = [torch.from_numpy(lin[f"lin{i}"]) for i in range(len(chns))]
        t0 = [torch.from_numpy(g[f"{net}_tap0_{i}"]) for i in range(len(chns))]
        t1 = [torch.from_numpy(g[f"{net}_tap1_{i}"]) for i in range(len(chns))]
        assert [t.shape[1] for t in t0] == chns
        total, vals = loss_ref.lpips_distance_ref(t0, t1, lins, per_layer=True)
        # reference quirk: `val = res[0]; val += res[l]` (networks_basic.py:85-87) accumulates IN PLACE, so the res[0] that
        # retPerLayer=True hands back is the total, not tap 0's term
        assert np.array_equal(g[f"{net}_res_0"], g[f"{net}_val"])
        for i, v in enumerate(vals):
            if i > 0:
                assert tuple(v.shape) == g[f"{net}_res_{i}"].shape and rel(v, g[f"{net}_res_{i}"]) < TOL, (net, i)
        assert rel(total, g[f"{net}_val"]) < TOL, net
        rest = sum(g[f"{net}_res_{i}"].astype(np.float64) for i in range(1, len(chns)))
        assert np.abs(vals[0].numpy() - (g[f"{net}_val"] - rest)).max() < 1e-6 * np.abs(g[f"{net}_val"]).max(), net
        assert rel(loss_ref.normalize_tensor_ref(t0[1]), g[f"{net}_unit0_1"]) < TOL
        assert float(loss_ref.normalize_tensor_ref(t0[1])[:, :, 0, 0].abs().max()) == 0.0        # all-zero pixel: 0 / (0 + eps)
    assert rel(loss_ref.scaling_layer_ref(torch.from_numpy(g["scale_in"])), g["scale_out"]) < TOL
    lin = golden("lpips_lin_squeeze.npz")
    lins = [torch.from_numpy(lin[f"lin{i}"]) for i in range(7)]
    bb = loss_ref.squeeze_backbone_random(0)
    a, b = torch.from_numpy(g["full_in0"]), torch.from_numpy(g["full_in1"])
    total, vals = loss_ref.lpips_ref(bb, lins, a, b, per_layer=True)
    assert tuple(total.shape) == (2, 1, 1, 1) and rel(total, g["full_val"]) < TOL
    assert rel(torch.stack([v.reshape(-1) for v in vals[1:]]), g["full_res"][1:]) < TOL and np.array_equal(g["full_res"][0], g["full_val"].reshape(-1))
    # PerceptualLoss.forward(normalize=True) maps [0,1] inputs to [-1,1] first (lpips/__init__.py:36-38)
    assert rel(loss_ref.lpips_ref(bb, lins, 2 * ((a + 1) / 2) - 1, 2 * ((b + 1) / 2) - 1), g["full_val_normalize"]) < TOL


@pytest.mark.parametrize("name", ["loop_tiny.npz", "loop_config0_256.npz"])
def test_every_step_sigma_and_latent_vs_reference_run(golden, name):
    """All 50 noise strengths and all 50 perturbed latents of the reference-driven runs (not only the selected step): the
    oracle's schedule and the package's host-side schedule table reproduce `latent_std * noise * ramp` (float32 tensor x python
    scalars, ...sqz_MSE.py:156) and `latent_in + randn * strength` (:71-73,157) bit for bit."""
    from morphganformer_amd.projection import noise_schedule
    g = golden(name)
    steps = g["eps"].shape[0]
    sig_pkg = noise_schedule(steps, float(g["latent_std"]), 0.05, 0.75)
    assert sig_pkg.dtype == np.float32
    mean = torch.from_numpy(g["latent_mean"])[None]
    for i in range(steps):
        s = loss_ref.noise_strength_ref(i / steps, float(g["latent_std"]))
        assert s == float(g["sigmas"][i]) and float(sig_pkg[i]) == float(g["sigmas"][i]), i
        lat = mean + torch.from_numpy(g["eps"][i]) * s
        assert np.array_equal(lat.numpy(), g["latents_n"][i]), i
        # the device kernel multiplies float32 eps by the float32 table entry and adds with two roundings: same numbers
        lat32 = (g["latent_mean"][None] + g["eps"][i] * sig_pkg[i]).astype(np.float32)
        assert np.array_equal(lat32, g["latents_n"][i]), i


def test_per_layer_ws_truncation_cutoff_and_att_tensor_vs_reference(golden):
    """Generator.forward's W+ / truncation_cutoff / return_att paths (networks.py:1304-1331, 935-941, 1222-1262) of the restatement
    against the reference module's outputs (tests/golden/wplus_tiny.npz)."""
    from oracle.generator_ref import list2tensor_ref, synthesis_ref, truncate_ref
    g = golden("wplus_tiny.npz")
    sd = to_torch_state(make_state_dict(TINY, 0))
    z, ws = torch.from_numpy(g["z"]), torch.from_numpy(g["ws"])
    taps = {}
    img = synthesis_ref(sd, ws, TINY, "const", taps=taps)
    assert rel(img, g["img_ws"]) < 1e-5 and rel(img, g["img_synthesis_subnet"]) < 1e-5
    order = [f"synthesis.b{res}.{name}" for res, name, *_rest, att, _nb in TINY.layer_table() if att]
    att = list2tensor_ref([taps[k + ":probs"] for k in order], TINY)
    assert tuple(att.shape) == tuple(g["att_shape"]) and rel(att[:, :, :, 0, 3::8, 5::8], g["att_sub"]) < 1e-5
    w = mapping_ref(sd, z, TINY)
    for psi, cut, kimg, kws in ((0.6, 5, "img_cut", "ws_cut"), (0.6, None, "img_psi", "ws_psi")):
        wst = truncate_ref(sd, w, TINY, psi, cut)
        assert rel(wst, g[kws]) < TOL
        assert rel(generator_ref(sd, z, TINY, "const", truncation_psi=psi, truncation_cutoff=cut), g[kimg]) < 1e-5
    assert float((torch.from_numpy(g["ws_cut"])[:, :, 5:] - w[:, :, None]).abs().max()) < 1e-6          # slots >= cutoff untouched
    # W+ gradient: the restatement's autograd with respect to per-layer latents against the reference module's (grad_ws)
    wg = ws.clone().requires_grad_(True)
    loss = synthesis_ref(sd, wg, TINY, "const").square().mean()
    (gws,) = torch.autograd.grad(loss, wg)
    assert abs(float(loss.detach()) - float(g["loss_ws"])) < 1e-5 * float(g["loss_ws"])
    assert rel(gws, g["grad_ws"]) < 1e-4


# ------------------------------------------------------------------------------------------------ OpenCV warp restatement (oracle/warp_ref.py)
def test_cv_warp_known_answers_by_hand():
    """Hand-computed KATs of the OpenCV rules oracle/warp_ref.py restates (no cv2 offline: these pin the RULES, not the library):
    a 1/64-pixel shift lands on the 1/32 grid by `+16 >> 5` (the tie goes UP), cvRound is half-to-even on the 1/1024 grid, the reflection is
    BORDER_REFLECT_101 at the PATCH border, fillConvexPoly's pixel set on an axis-aligned right triangle is x + y <= 4 (the last scanline
    comes from the outline only), np.uint8 truncates."""
    from oracle import warp_ref as W
    ramp = (32.0 * np.arange(8, dtype=np.float32))[None, :, None].repeat(3, 0).repeat(1, 2)        # [3, 8, 1]: S[x] = 32 x
    # src -> dst translation by -1/64 px: dst x reads src x + 1/64 -> fixed point 16/1024 -> (16 + 16) >> 5 = 1/32 -> 32 x + 1 exactly
    out = W.warp_affine_linear_reflect101(ramp, np.array([[1, 0, -1.0 / 64], [0, 1, 0]]), (6, 3))
    assert np.array_equal(out[1, :, 0], 32.0 * np.arange(6) + 1)
    # 15/1024 stays below the tie: (15 + 16) >> 5 = 0 -> the pixel itself
    out = W.warp_affine_linear_reflect101(ramp, np.array([[1, 0, -15.0 / 1024], [0, 1, 0]]), (6, 3))
    assert np.array_equal(out[1, :, 0], 32.0 * np.arange(6))
    # cvRound half-to-even on the 1/1024 grid: 16.5/1024 -> 16 (tie up to 1/32), 15.5/1024 -> 16 as well (even), 14.5 -> 14 (stays)
    for shift, want in ((16.5, 1.0), (15.5, 1.0), (14.5, 0.0)):
        out = W.warp_affine_linear_reflect101(ramp, np.array([[1, 0, -shift / 1024], [0, 1, 0]]), (2, 1))
        assert out[0, 0, 0] == want, (shift, out[0, 0, 0])
    assert W.cv_round(0.5) == 0 and W.cv_round(1.5) == 2 and W.cv_round(2.5) == 2 and W.cv_round(-0.5) == 0
    # BORDER_REFLECT_101 at the patch: a 4-wide patch [10 20 30 40], samples at 3.5 and -0.5
    patch = np.array([10, 20, 30, 40], np.float32)[None, :, None]
    out = W.warp_affine_linear_reflect101(patch, np.array([[1, 0, -3.5], [0, 1, 0]]), (1, 1))
    assert out[0, 0, 0] == 35.0                                            # 40 * 1/2 + reflect(4) = 30 * 1/2
    out = W.warp_affine_linear_reflect101(patch, np.array([[1, 0, 0.5], [0, 1, 0]]), (1, 1))
    assert out[0, 0, 0] == 15.0                                            # reflect(-1) = 20 * 1/2 + 10 * 1/2
    assert [W.border_reflect_101(p, 4) for p in (-2, -1, 0, 3, 4, 5, 7)] == [2, 1, 0, 3, 2, 1, 1] and W.border_reflect_101(9, 1) == 0
    # fillConvexPoly: right triangle (0,0) (4,0) (0,4)
    m = W.fill_convex_poly(6, 6, [(0, 0), (4, 0), (0, 4)])
    yy, xx = np.mgrid[0:6, 0:6]
    assert np.array_equal(m, xx + yy <= 4)
    # Bresenham, left to right, tie rule: (0,0) -> (4,2) visits y = 0 0 1 1 2?  minor steps when the ideal exceeds by MORE than 1/2: k=1: 0.5 -> 0
    assert W.line8((0, 0), (4, 2)) == [(0, 0), (1, 0), (2, 1), (3, 1), (4, 2)]
    assert W.line8((4, 2), (0, 0)) == W.line8((0, 0), (4, 2))             # left_to_right: the same pixels from either end
    assert W.line8((0, 0), (2, -4)) == [(0, 0), (0, -1), (1, -2), (1, -3), (2, -4)]
    assert W.bounding_rect_f32([(10.5, 3.0), (12.0, 7.5), (11.25, 4.0)]) == (10, 3, 3, 5)
    # getAffineTransform + inversion: a pure scale-and-shift has a closed form
    M = W.get_affine_transform([(0, 0), (4, 0), (0, 4)], [(1, 2), (9, 2), (1, 10)])
    assert np.allclose(M, [[2, 0, 1], [0, 2, 2]], atol=1e-15)
    assert np.allclose(W.invert_affine(M), [0.5, 0, -0.5, 0, 0.5, -1.0], atol=1e-15)


def test_cv_warp_host_plan_equals_the_literal_transcription():
    """The product's closed-form host set-up (drivers: Bresenham by formula, scanline runs between vertex rows, row-wise LU) against the
    oracle's literal one-step-at-a-time transcription: identical pixel sets, bit-identical matrices, on random and degenerate triangles;
    and the oracle's vectorised patch warp equals its literal per-pixel loop bit for bit."""
    from morphganformer_amd import drivers as D
    from oracle import warp_ref as W
    rng = np.random.Generator(np.random.PCG64(11))
    for a in range(-6, 7):
        for b in range(-6, 7):
            xs, ys = D._cv_line8((3, 2), (3 + a, 2 + b))
            assert list(zip(xs.tolist(), ys.tolist())) == W.line8((3, 2), (3 + a, 2 + b)), (a, b)
    tris = [rng.integers(0, 40, (3, 2)) for _ in range(300)]
    tris += [np.array(t) for t in ([(0, 0), (9, 0), (4, 0)], [(5, 5), (5, 5), (5, 5)], [(0, 3), (7, 3), (2, 9)], [(2, 0), (2, 8), (6, 8)],
                                    [(0, 0), (39, 39), (0, 39)], [(10, 1), (11, 30), (12, 2)])]
    for t in tris:
        assert np.array_equal(D._cv_fill_convex_poly(40, 40, t), W.fill_convex_poly(40, 40, t)), t.tolist()
    for _ in range(200):
        s, d = rng.uniform(0, 200, (3, 2)), rng.uniform(0, 200, (3, 2))
        if abs(np.linalg.det(np.c_[np.float32(s), np.ones(3)])) < 1.0:
            continue
        want = W.invert_affine(W.get_affine_transform(np.float32(s), np.float32(d)))
        assert np.array_equal(D._cv_affine_inverse(np.float32(s), np.float32(d)), want)
    # a collinear SOURCE triangle: LU64f reports "singular", cv::solve zeroes X, getAffineTransform returns the zero matrix without
    # looking, warpAffine's inverse of it is zero too -- every pixel of the patch then samples source pixel (0, 0)
    for s in ([(0, 22), (0, 0), (0, 20)], [(1, 1), (3, 3), (7, 7)], [(4, 4), (4, 4), (9, 2)]):
        d = [(0, 15), (2, 0), (4, 19)]
        assert np.array_equal(W.get_affine_transform(np.float32(s), np.float32(d)), np.zeros((2, 3)))
        assert np.array_equal(D._cv_affine_inverse(np.float32(s), np.float32(d)), np.zeros(6))
        assert np.array_equal(W.invert_affine(np.zeros((2, 3))), np.zeros(6))
    assert D._cv_bounding_rect([(10.5, 3.0), (12.0, 7.5), (11.25, 4.0)]) == (10, 3, 3, 5)
    src = rng.integers(0, 256, (9, 11, 3)).astype(np.float32)
    for _ in range(20):
        M = np.array([[1, 0, 0], [0, 1, 0]], np.float64) + rng.uniform(-0.6, 0.6, (2, 3)) * np.array([1, 1, 6.0])
        assert np.array_equal(W.warp_affine_linear_reflect101(src, M, (13, 10)), W.warp_affine_linear_reflect101_rows(src, M, (13, 10)))


def test_ssim_ref_known_answers_by_hand():
    """oracle.loss_ref.ssim_ref restates skimage's compare_ssim (absent here, so unpinned against the library): held by answers that follow
    from the published formula -- identical images 1; two constant images (2ab + C1) / (a^2 + b^2 + C1) (both variances and the covariance
    vanish, so the contrast term is C2 / C2); a window-by-window evaluation with numpy's own sample moments on a small random pair; a
    1-D (flattened) input is refused like compare_ssim refuses it (what 1024_example_SSIM.py:158 runs into)."""
    from oracle.loss_ref import dssim_ref, ssim_ref, to_u8_ref
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (13, 17, 3)).astype(np.uint8)
    b = rng.integers(0, 256, (13, 17, 3)).astype(np.uint8)
    assert ssim_ref(a, a) == 1.0
    x, y = np.full((9, 11, 1), 100, np.uint8), np.full((9, 11, 1), 50, np.uint8)
    c1 = (0.01 * 255) ** 2
    assert abs(ssim_ref(x, y) - (2 * 100 * 50 + c1) / (100 ** 2 + 50 ** 2 + c1)) < 1e-14
    c2 = (0.03 * 255) ** 2
    per = []
    for ch in range(3):
        vals = []
        for i in range(13 - 6):
            for j in range(17 - 6):
                p, q = a[i:i + 7, j:j + 7, ch].astype(np.float64).ravel(), b[i:i + 7, j:j + 7, ch].astype(np.float64).ravel()
                cov = np.cov(p, q, ddof=1)
                vals.append((2 * p.mean() * q.mean() + c1) * (2 * cov[0, 1] + c2) / ((p.mean() ** 2 + q.mean() ** 2 + c1) * (cov[0, 0] + cov[1, 1] + c2)))
        per.append(np.mean(vals))
    assert abs(ssim_ref(a, b) - np.mean(per)) < 1e-12
    with pytest.raises((ValueError, AssertionError)):
        ssim_ref(a.reshape(-1), b.reshape(-1))
    # the quantisation of the saved image: rint, not truncation (misc.py:115-116)
    assert to_u8_ref(np.array([-1.0, -0.9961, 0.0, 0.0039, 1.0, 1.5], np.float32)).tolist() == [0, 0, 128, 128, 255, 255]
    f = (a.transpose(2, 0, 1).astype(np.float32) / 127.5 - 1)
    assert dssim_ref(f, f) == 0 and dssim_ref(f, f).dtype == np.float32


def test_lbp_ref_known_answers_by_hand():
    """oracle.loss_ref restates OpenCV's BGR2GRAY / INTER_LINEAR resize and skimage's local_binary_pattern(24, 3, 'uniform') (both absent:
    unpinned against the libraries).  Held by answers that follow from the published algorithms: a flat image -- every interior sample
    equals the centre or misses it by an ulp, so the interior code is whatever a SCALAR walk through skimage's expressions gives, the corner
    sees the 7 samples with non-negative offsets, an edge 13; a bright pixel on black is code 0, a black pixel on white 24; the vectorised
    restatement equals that scalar walk on a random image; gray weights and the two resize implementations (product host code, oracle)."""
    from oracle.loss_ref import cv_bgr2gray_u8_ref, cv_resize_linear_gray_ref, lbp_cosine_distance_ref, lbp_uniform_ref
    from morphganformer_amd.drivers import cv_resize_linear_u8
    P, R = 24, 3
    ang = 2 * np.pi * np.arange(P, dtype=np.double) / P
    rp, cp = np.round(-R * np.sin(ang), 5), np.round(R * np.cos(ang), 5)

    def scalar(img):                       # _texture.pyx line by line, one pixel at a time
        rows, cols = img.shape
        f = img.astype(np.double)
        out = np.zeros((rows, cols))
        px = lambda r, c: float(f[r, c]) if (0 <= r < rows and 0 <= c < cols) else 0.0
        for r in range(rows):
            for c in range(cols):
                s = []
                for p in range(P):
                    y, x = r + float(rp[p]), c + float(cp[p])
                    minr, minc, maxr, maxc = math.floor(y), math.floor(x), math.ceil(y), math.ceil(x)
                    dr, dc = y - minr, x - minc
                    top = (1 - dc) * px(minr, minc) + dc * px(minr, maxc)
                    bottom = (1 - dc) * px(maxr, minc) + dc * px(maxr, maxc)
                    s.append(1 if ((1 - dr) * top + dr * bottom) - float(f[r, c]) >= 0 else 0)
                changes = sum(s[i] != s[i + 1] for i in range(P - 1))
                out[r, c] = sum(s) if changes <= 2 else P + 1
        return out

    flat = np.full((12, 13), 64, np.uint8)                      # 64 is a power of two: (1 - dc) * 64 + dc * 64 == 64 exactly
    got = lbp_uniform_ref(flat)
    assert got[6, 6] == 24 and got[0, 0] == 7 and got[0, 6] == 13 and got[11, 12] == 7
    spot = np.zeros((11, 11), np.uint8); spot[5, 5] = 200
    assert lbp_uniform_ref(spot)[5, 5] == 0
    hole = np.full((11, 11), 255, np.uint8); hole[5, 5] = 0
    assert lbp_uniform_ref(hole)[5, 5] == 24
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (14, 15)).astype(np.uint8)
    img[3:9, 4:11] = 77                                          # a flat patch: the ulp-level ties of the interpolation are part of the answer
    assert np.array_equal(lbp_uniform_ref(img), scalar(img))
    v = lbp_uniform_ref(img)
    assert v.dtype == np.float64 and v.min() >= 0 and v.max() <= 25
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    assert cv_bgr2gray_u8_ref(rgb, blue_first=True).tolist() == [[29, 150, 76, (10 * 1868 + 20 * 9617 + 30 * 4899 + 8192) >> 14]]
    assert cv_bgr2gray_u8_ref(rgb, blue_first=False).tolist() == [[76, 150, 29, (30 * 1868 + 20 * 9617 + 10 * 4899 + 8192) >> 14]]
    g = rng.integers(0, 256, (61, 47)).astype(np.uint8)
    for (w, h) in ((224, 224), (30, 20), (47, 61)):
        assert np.array_equal(cv_resize_linear_gray_ref(g, w, h), cv_resize_linear_u8(g[..., None], w, h)[..., 0])
    assert abs(lbp_cosine_distance_ref(v, v)) < 1e-15 and abs(lbp_cosine_distance_ref(v, 2 * v)) < 1e-15
    assert abs(lbp_cosine_distance_ref([1, 0], [0, 1]) - 1) < 1e-15
