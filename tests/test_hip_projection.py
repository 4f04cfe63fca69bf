"""GPU parity of the loss kernels and of the literal-mode projection loop.

 * LPIPS(squeeze): HIP feature extractor + distance vs the CPU oracle (torch conv2d/max_pool2d) on seeded backbone
   weights and the reference's vendored lin heads; 1e-3 relative (float32 re-association through 26 conv layers).
 * Wing / MSE: vs the reference's own KATs (tests/golden/loss_kats.npz).
 * Loop: against tests/golden/loop_tiny.npz, which was produced by driving the REFERENCE Generator and the REFERENCE
   WingLoss through the loop of 1024_example_wing_loss_perceptual_sqz_MSE.py:152-189 with injected noise/landmarks:
   best step exact (integer), best latent bit-exact, losses to 1e-3.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_wing_and_mse_kats(golden):
    from morphganformer_amd import _lib
    from morphganformer_amd.wing_loss import WingLoss
    g = golden("loss_kats.npz")
    w = WingLoss()
    v = w(torch.zeros(2, 68, 64, 64).cuda(), torch.ones(2, 68, 64, 64).cuda())
    assert v.dtype == torch.float64 and abs(float(v) - float(g["wing_ones_zeros"])) < 3e-6     # the KAT is float32 in the reference; the kernel is float64
    v = w(torch.from_numpy(g["wing_small_pred"]).cuda(), torch.from_numpy(g["wing_small_target"]).cuda())
    assert abs(float(v) - float(g["wing_small"])) < 1e-12
    v = w(torch.from_numpy(g["wing_rand_pred"]).cuda(), torch.from_numpy(g["wing_rand_target"]).cuda())
    assert abs(float(v) - float(g["wing_rand"])) < 1e-12 * abs(float(g["wing_rand"])) + 1e-12
    # AdaptiveWingLoss: the reference's own self-test value + random heat maps vs the oracle
    from morphganformer_amd.wing_loss import AdaptiveWingLoss
    from oracle.loss_ref import adaptive_wing_loss_ref
    aw = AdaptiveWingLoss()
    v = aw(torch.zeros(68, 2).cuda(), torch.ones(68, 2).cuda())
    assert abs(float(v) - float(g["awing_ones_zeros"])) < 3e-6                       # float32 KAT
    torch.manual_seed(1)
    hp, ht = torch.rand(2, 5, 16, 16, dtype=torch.float64), torch.rand(2, 5, 16, 16, dtype=torch.float64)
    assert abs(float(aw(hp.cuda(), ht.cuda())) - float(adaptive_wing_loss_ref(hp, ht))) < 1e-12
    torch.manual_seed(0)
    a, b = torch.randn(1, 3, 129, 67), torch.randn(1, 3, 129, 67)
    out = torch.zeros(1).cuda()
    scratch = torch.empty(int(_lib.lib().mgf_reduce_scratch_floats())).cuda()
    ad, bd = a.cuda(), b.cuda()
    _lib.check(_lib.lib().mgf_mse_f32(out.data_ptr(), ad.data_ptr(), bd.data_ptr(), 1, a.numel(), 0, 1.0, 0, scratch.data_ptr(), _lib.stream_ptr()))
    ref = float(torch.nn.functional.mse_loss(a.double(), b.double()))
    assert abs(float(out) - ref) < 1e-6 * ref


@pytest.mark.parametrize("res", [64, 131])
def test_lpips_squeeze_vs_oracle(res):
    from morphganformer_amd.lpips import PerceptualLoss, random_squeeze_backbone
    from oracle.loss_ref import lpips_ref, squeeze_backbone_random, squeeze_features_ref, LPIPS_SHIFT, LPIPS_SCALE
    torch.manual_seed(res)
    x0 = (torch.rand(1, 3, res, res) * 2 - 1)
    x1 = (x0 + 0.3 * torch.randn(1, 3, res, res)).clamp(-1, 1)
    bb_np = random_squeeze_backbone(0)
    bb = squeeze_backbone_random(0)
    for k in bb:
        assert np.array_equal(bb[k].numpy(), bb_np[k]), k          # package and oracle generate identical weights
    P = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, backbone_state=bb_np)
    lins = [l.cpu() for l in P.lins]
    ref_total, ref_layers = lpips_ref(bb, lins, x0, x1, per_layer=True)
    out = P(x0.cuda(), x1.cuda())
    assert tuple(out.shape) == (1, 1, 1, 1)
    assert abs(float(out) - float(ref_total)) < 1e-3 * abs(float(ref_total)), (float(out), float(ref_total))
    # feature taps individually
    shift = torch.tensor(LPIPS_SHIFT).reshape(1, 3, 1, 1)
    scale = torch.tensor(LPIPS_SCALE).reshape(1, 3, 1, 1)
    ref_taps = squeeze_features_ref(bb, (x0 - shift) / scale)
    taps = P._features(1, res, res)(x0.cuda())
    assert len(taps) == 7
    for i, (t, r) in enumerate(zip(taps, ref_taps)):
        assert tuple(t.shape) == tuple(r.shape), i
        assert float((t.cpu() - r).abs().max() / r.abs().max()) < 1e-4, i
    # identical images -> exactly zero
    assert float(P(x0.cuda(), x0.cuda())) == 0.0


@pytest.mark.parametrize("res", [64, 130, 131, 1024])
def test_lpips_fused_stem_vs_oracle(res):
    """The one-pass stem (conv 3->64 s2 + ReLU + tap-0 distance + ceil-mode max-pool, csrc/lpips_stem.hip) against the oracle's
    conv2d / max_pool2d / normalise: pooled map, the normalised reference tap, the tap-0 distance of a batch of 3 images,
    and agreement of the whole LPIPS value with the unfused kernels."""
    from morphganformer_amd import _lib
    from morphganformer_amd.lpips import PerceptualLoss, random_squeeze_backbone
    from oracle.loss_ref import squeeze_backbone_random, LPIPS_SHIFT, LPIPS_SCALE
    import torch.nn.functional as F
    torch.manual_seed(res)
    n = 3
    x = (torch.rand(n, 3, res, res) * 2 - 1)
    tgt = (x[:1] + 0.3 * torch.randn(1, 3, res, res)).clamp(-1, 1)
    bb = squeeze_backbone_random(0)
    P = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, backbone_state=random_squeeze_backbone(0))
    assert P.fused_stem
    shift = torch.tensor(LPIPS_SHIFT).reshape(1, 3, 1, 1)
    scale = torch.tensor(LPIPS_SCALE).reshape(1, 3, 1, 1)
    w0, b0 = bb["features.0.weight"].double(), bb["features.0.bias"].double()

    def tap0(img):
        return F.relu(F.conv2d(((img - shift) / scale).double(), w0, b0, stride=2))

    def unit(t):
        return t / (t.square().sum(1, keepdim=True).sqrt() + 1e-10)

    f = P._features(n, res, res)
    # reference mode: normalised tap 0 + pooled map
    feat = torch.empty(n, 64, *f.shapes[1][1:], device="cuda")
    pooled = f.stem(x.cuda(), feat_out=feat).clone()
    r0 = tap0(x)
    rp = F.max_pool2d(r0, 3, 2, ceil_mode=True)
    assert tuple(pooled.shape) == tuple(rp.shape)
    assert float((pooled.cpu() - rp).abs().max()) < 2e-5 * float(rp.abs().max())
    assert float((feat.cpu() - unit(r0)).abs().max()) < 2e-5
    # distance mode against the (single) target's normalised tap 0
    f1 = P._features(1, res, res)
    tfeat = torch.empty(1, 64, *f1.shapes[1][1:], device="cuda")
    f1.stem(tgt.cuda(), feat_out=tfeat)
    out = torch.full([n], 7.0, device="cuda")
    scratch = torch.empty(n * int(_lib.lib().mgf_reduce_scratch_floats()), device="cuda")
    pooled2 = f.stem(x.cuda(), feat_ref=tfeat, lin=P.lins[0], dist_out=out, scratch=scratch)
    assert torch.equal(pooled2, pooled)
    lin0 = P.lins[0].cpu().double().reshape(1, 64, 1, 1)
    want = ((unit(r0) - unit(tap0(tgt))).square() * lin0).sum(1).mean(dim=(1, 2))
    assert float((out.cpu().double() - want).abs().max()) < 1e-4 * float(want.abs().max())
    # whole-LPIPS agreement with the separate conv / pool / distance kernels, and the exact zero on identical images
    P.set_target(tgt.cuda())
    fused = torch.zeros(n, device="cuda")
    P.distance_into(fused, x.cuda())
    Q = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, backbone_state=random_squeeze_backbone(0))
    Q.fused_stem = False
    Q.set_target(tgt.cuda())
    plain = torch.zeros(n, device="cuda")
    Q.distance_into(plain, x.cuda())
    assert float((fused - plain).abs().max()) < 1e-4 * float(plain.abs().max())
    assert float(P(tgt.cuda(), tgt.cuda())) == 0.0


@pytest.mark.parametrize("net,res", [("vgg", 64), ("vgg", 75), ("alex", 96), ("alex", 131)])
def test_lpips_vgg_alex_vs_oracle(net, res):
    """The other two backbones of lpips.PerceptualLoss (pretrained_networks.py:58-135): taps and distance vs the oracle, seeded
    random backbone weights + the reference's vendored lin heads.  AlexNet's 11x11/5x5 convs run as chained <=9-tap launches."""
    from morphganformer_amd.lpips import PerceptualLoss, random_backbone
    from oracle.loss_ref import backbone_random, lpips_ref, sequential_features_ref, LPIPS_SHIFT, LPIPS_SCALE
    torch.manual_seed(res)
    x0 = (torch.rand(2, 3, res, res) * 2 - 1)
    x1 = (x0[:1] + 0.3 * torch.randn(1, 3, res, res)).clamp(-1, 1)
    bb_np, bb = random_backbone(net, 0), backbone_random(net, 0)
    for k in bb:
        assert np.array_equal(bb[k].numpy(), bb_np[k]), k
    P = PerceptualLoss(model="net-lin", net=net, use_gpu=True, backbone_state=bb_np)
    assert len(P.lins) == 5
    lins = [l.cpu() for l in P.lins]
    shift = torch.tensor(LPIPS_SHIFT).reshape(1, 3, 1, 1)
    scale = torch.tensor(LPIPS_SCALE).reshape(1, 3, 1, 1)
    ref_taps = sequential_features_ref(net, bb, (x0 - shift) / scale)
    taps = P._features(2, res, res)(x0.cuda())
    for i, (t, r) in enumerate(zip(taps, ref_taps)):
        assert tuple(t.shape) == tuple(r.shape), i
        assert float((t.cpu() - r).abs().max() / r.abs().max()) < 1e-4, i
    P.set_target(x1.cuda())
    out = torch.zeros(2, device="cuda")
    P.distance_into(out, x0.cuda())
    for i in range(2):
        ref = float(lpips_ref(bb, lins, x0[i:i + 1], x1, net=net))
        assert abs(float(out[i]) - ref) < 1e-3 * abs(ref), (i, float(out[i]), ref)
    assert float(P(x1.cuda(), x1.cuda())) == 0.0


def _engine_from_golden(g, use_graph, steps=None, batch=1):
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
    steps = steps or int(g["eps"].shape[0])
    args = ProjectionArgs(step=int(g["eps"].shape[0]))
    eng = ProjectionEngine(G, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]),
                           args, percept=None, use_mse=True, lm_target=g["lm_target"], lm_steps=g["lm_steps"],
                           eps=torch.from_numpy(g["eps"]).cuda(), noise_mode="const", use_graph=use_graph, batch=batch)
    return eng


@pytest.mark.parametrize("use_graph", [False, True])
def test_literal_loop_matches_reference_run(golden, use_graph):
    g = golden("loop_tiny.npz")
    eng = _engine_from_golden(g, use_graph)
    lat, bstep, bloss, losses = eng.run().result()
    assert bstep == int(g["best_step"])                                       # integer output: exact
    assert np.array_equal(lat.numpy(), g["best_latent"])                      # latent: bit-exact under injected noise
    assert np.abs(losses - g["losses"]).max() < 1e-3 * np.abs(g["losses"]).max()
    assert abs(bloss - float(g["best_loss"])) < 1e-3 * float(g["best_loss"])
    # idempotence: a second engine on the same inputs reproduces the loss history bit for bit (deterministic reductions)
    lat2, bstep2, bloss2, losses2 = _engine_from_golden(g, use_graph).run().result()
    assert bstep2 == bstep and bloss2 == bloss and np.array_equal(losses2, losses)


@pytest.mark.parametrize("batch", [2, 4, 7])
def test_batched_steps_equal_sequential_loop(golden, batch):
    """Evaluating `batch` loop steps per generator forward (in-order selection) reproduces the sequential reference run:
    same best step, bit-identical best latent, same loss history; 50 steps with batch 4 / 7 also covers a ragged last batch."""
    g = golden("loop_tiny.npz")
    lat, bstep, bloss, losses = _engine_from_golden(g, True, batch=batch).run().result()
    assert bstep == int(g["best_step"])
    assert np.array_equal(lat.numpy(), g["best_latent"])
    assert not np.isnan(losses).any()
    assert np.abs(losses - g["losses"]).max() < 1e-3 * np.abs(g["losses"]).max()
    lat1, bstep1, bloss1, losses1 = _engine_from_golden(g, False, batch=1).run().result()
    assert bstep1 == bstep and np.abs(losses1 - losses).max() < 1e-5 * np.abs(losses).max()


def test_loop_no_face_and_never_improves(golden):
    from morphganformer_amd.projection import ProjectionArgs
    g = golden("loop_tiny.npz")
    steps = int(g["eps"].shape[0])
    # "no face detected" on every step except 3 and 7 -> best step must be one of them; skipped steps record NaN
    eng = _engine_from_golden(g, False)
    valid = np.zeros(steps, np.int32); valid[[3, 7]] = 1
    eng.valid = torch.from_numpy(valid).cuda()
    lat, bstep, bloss, losses = eng.run().result()
    want = 3 if g["losses"][3] <= g["losses"][7] else 7
    assert bstep == want and np.isnan(losses[0]) and not np.isnan(losses[3])
    # min_loss_init below every loss -> the reference raises IndexError (latent_path[-1] on an empty list, :208)
    eng = _engine_from_golden(g, False)
    eng.min_loss.fill_(1e-9)
    with pytest.raises(IndexError):
        eng.run().result()


def test_full_size_loop_properties():
    """BASELINE full size (1024^2): a short literal run with all three losses; size-independent properties:
    best loss == min of the recorded history, best step == argmin, the stored latent regenerates (const noise) an image
    whose MSE+LPIPS equals what a fresh evaluation gives, and the history is reproducible under the same seeds."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, latent_stats, synthetic_landmarks
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    G = Generator(make_state_dict(FULL1024, seed=0), FULL1024, "cuda", max_batch=1)
    target = G(torch.from_numpy(synthetic_latents(FULL1024, 1, 1000)).cuda(), None, noise_mode="const")[0].clamp(-1, 1).clone()
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    mean, std = latent_stats(G, 10000, "cuda", gen)
    assert abs(float(std) - 23.32) < 0.2                         # SURVEY.md 8c KAT (seeded differently: loose)
    steps = 12
    lm_t, lm_s = synthetic_landmarks(steps, 1024, 7)
    hist = []
    for _ in range(2):
        P = PerceptualLoss(net="squeeze")
        eng = ProjectionEngine(G, target, mean, std, ProjectionArgs(step=steps), percept=P, lm_target=lm_t, lm_steps=lm_s,
                               noise_mode="const", seed=5, use_graph=True)
        lat, bstep, bloss, losses = eng.run().result()
        hist.append(losses)
        assert not np.isnan(losses).any()
        assert bstep == int(np.argmin(losses)) and bloss == float(losses.min())
    assert np.array_equal(hist[0], hist[1])
    # the selected latent really is eps[bstep]*sigma[bstep] + mean
    expect = eng.latent_in + eng.eps[bstep] * eng.sigma[bstep]
    assert torch.equal(lat.cuda(), expect)


def test_percept_mse_objective_variant(golden):
    """1024_example_percept_MSE.py:147: total = 0.5 * LPIPS(vgg) + 0.5 * MSE -- coefficient on the perceptual term and the VGG backbone
    inside the loop; every recorded loss equals the separately evaluated terms."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    g = golden("loop_tiny.npz")
    G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
    steps = 4
    tgt = torch.from_numpy(g["target"]).cuda()
    P = PerceptualLoss(model="net-lin", net="vgg", use_gpu=True)
    eng = ProjectionEngine(G, tgt, torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]),
                           ProjectionArgs(step=steps, percept_weight=0.5, beta=0.5), percept=P, use_mse=True,
                           eps=torch.from_numpy(g["eps"][:steps]).cuda(), noise_mode="const", batch=2)
    lat, bstep, bloss, losses = eng.run().result()
    for i in range(steps):
        sigma = np.float32(float(g["latent_std"]) * 0.05 * max(0, 1 - (i / steps) / 0.75) ** 2)
        z = torch.from_numpy(g["latent_mean"])[None] + torch.from_numpy(g["eps"][i]) * float(sigma)
        img = G(z.cuda(), None, noise_mode="const")[0]
        want = 0.5 * float(P(img, tgt)) + 0.5 * float(torch.nn.functional.mse_loss(img, tgt))
        assert abs(losses[i] - want) < 1e-5 * abs(want), (i, losses[i], want)
    assert bstep == int(np.argmin(losses))


def test_adaptive_wing_objective_variant(golden):
    """1024_example_wing_loss_adaptive.py: best-of selection on lamda * AdaptiveWing(landmarks) alone (normalised landmark
    coordinates so that the exponent alpha - y stays in the loss's working range)."""
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from oracle.loss_ref import adaptive_wing_loss_ref
    g = golden("loop_tiny.npz")
    steps = 9
    lm_t, lm_s = g["lm_target"] / 64.0, g["lm_steps"][:steps] / 64.0
    eng = ProjectionEngine(_tiny_gen(), torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(),
                           float(g["latent_std"]), ProjectionArgs(step=steps, lamda=1e-5, min_loss_init=1e5), percept=None, use_mse=False,
                           lm_target=lm_t, lm_steps=lm_s, eps=torch.from_numpy(g["eps"][:steps]).cuda(), noise_mode="const", batch=4,
                           wing_kind="awing")
    lat, bstep, bloss, losses = eng.run().result()
    want = np.array([1e-5 * float(adaptive_wing_loss_ref(torch.from_numpy(lm_s[i]), torch.from_numpy(lm_t))) for i in range(steps)])
    assert np.allclose(losses, want, rtol=1e-12) and bstep == int(np.argmin(want))


def _tiny_gen():
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    return Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)


@pytest.mark.parametrize("use_graph", [False, True])
def test_pipelined_mode_equals_plain_loop(golden, use_graph):
    """Losses of batch i on a side stream while the generator already runs batch i+1: same best step, bit-identical latent and
    loss history as the plain loop (and hence as the reference run), also across several run() calls and a ragged last batch."""
    g = golden("loop_tiny.npz")
    eng = _engine_from_golden(g, use_graph, batch=4)
    eng2 = _engine_from_golden(g, use_graph, batch=4)
    eng2.__init__(eng2.G, eng2.target, torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]), eng2.args, percept=None,
                  use_mse=True, lm_target=g["lm_target"], lm_steps=g["lm_steps"], eps=torch.from_numpy(g["eps"]).cuda(),
                  noise_mode="const", use_graph=use_graph, batch=4, pipeline=True)
    lat, bstep, bloss, losses = eng.run().result()
    eng2.run(12)
    eng2.run(38)                                                          # 50 steps in two calls; 50 = 12 * 4 + 2 (ragged tail)
    lat2, bstep2, bloss2, losses2 = eng2.result()
    assert bstep2 == bstep == int(g["best_step"]) and torch.equal(lat2, lat) and bloss2 == bloss
    assert np.array_equal(losses2, losses)
