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
    # min_loss_init below every loss -> the reference raises IndexError (latent_path[-1] on an empty list, :224)
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
        P = PerceptualLoss(net="squeeze", allow_random_backbone=True)
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


def test_full_size_loop_bench_batch_equals_batch1():
    """configs[1] as bench.py runs it -- 1024^2, Wing + LPIPS(squeeze) + MSE, drivers.DEFAULT_BATCH (32) loop steps per generator forward
    -- against the same 70 steps (two full launch sequences and a ragged third) evaluated one per forward, on an injected eps stream and constant per-layer noise: same best step, bit-identical best
    latent, loss history within 1e-5 (the two batch sizes take different kernel shapes; the selection is in step order either way)."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, latent_stats, synthetic_landmarks
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    G = Generator(make_state_dict(FULL1024, seed=0), FULL1024, "cuda", max_batch=1)
    target = G(torch.from_numpy(synthetic_latents(FULL1024, 1, 1000)).cuda(), None, noise_mode="const")[0].clamp(-1, 1).clone()
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    mean, std = latent_stats(G, 10000, "cuda", gen)
    from morphganformer_amd.drivers import DEFAULT_BATCH
    steps = 70
    lm_t, lm_s = synthetic_landmarks(steps, 1024, 7)
    eps = torch.randn(steps, 1, FULL1024.k, FULL1024.z_dim, device="cuda", generator=gen)
    out = {}
    for batch in (DEFAULT_BATCH, 1):
        P = PerceptualLoss(net="squeeze", allow_random_backbone=True)
        eng = ProjectionEngine(G, target, mean, std, ProjectionArgs(step=steps), percept=P, use_mse=True, lm_target=lm_t, lm_steps=lm_s,
                               eps=eps, noise_mode="const", use_graph=True, batch=batch)
        out[batch] = eng.run().result()
    (lat_a, step_a, loss_a, hist_a), (lat_b, step_b, loss_b, hist_b) = out[DEFAULT_BATCH], out[1]
    assert not np.isnan(hist_a).any() and not np.isnan(hist_b).any()
    assert step_a == step_b == int(np.argmin(hist_b))
    assert torch.equal(lat_a, lat_b)
    assert np.abs(hist_a - hist_b).max() <= 1e-5 * np.abs(hist_b).max()
    assert abs(loss_a - loss_b) <= 1e-5 * abs(loss_b)


def test_full_size_loop_64_candidates_per_forward():
    """64 loop steps per generator forward at 1024^2: the blur's input [64, 32, 1025, 1025] exceeds the 2^31 elements of the plug-in
    contract (upfirdn2d.cpp:14-15), which the engine-internal FIR path used to inherit -- a silent wall at 32.  The
    engine's own calls carry MGF_FILTER_LARGE now; the run equals the 32-per-forward run: same best step, bit-identical latent, history to
    1e-5.  The plug-in entry still refuses such a tensor."""
    from morphganformer_amd import _lib
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, latent_stats
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    from morphganformer_amd.torch_utils.ops import upfirdn2d
    G = Generator(make_state_dict(FULL1024, seed=0), FULL1024, "cuda", max_batch=1)
    target = G(torch.from_numpy(synthetic_latents(FULL1024, 1, 1000)).cuda(), None, noise_mode="const")[0].clamp(-1, 1).clone()
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    mean, std = latent_stats(G, 10000, "cuda", gen)
    steps = 64
    eps = torch.randn(steps, 1, FULL1024.k, FULL1024.z_dim, device="cuda", generator=gen)
    out = {}
    for batch in (64, 32):
        eng = ProjectionEngine(G, target, mean, std, ProjectionArgs(step=steps), percept=None, use_mse=True, eps=eps, noise_mode="const",
                               use_graph=False, batch=batch)
        out[batch] = eng.run().result()
        del eng
    (lat_a, step_a, loss_a, hist_a), (lat_b, step_b, loss_b, hist_b) = out[64], out[32]
    assert not np.isnan(hist_a).any()
    assert step_a == step_b and torch.equal(lat_a, lat_b)
    assert np.abs(hist_a - hist_b).max() <= 1e-5 * np.abs(hist_b).max()
    big = torch.empty(33, 64, 1025, 1025, device="cuda")                # 2.2e9 elements through the reference-signature operator
    with pytest.raises((_lib.MgfError, RuntimeError), match="too large"):
        upfirdn2d.upfirdn2d(big, upfirdn2d.setup_filter([1, 3, 3, 1], device="cuda"), padding=1)


def test_full_size_loop_vs_oracle():
    """configs[1] at FULL size against the CPU oracle, not against itself: 1024^2, Wing + LPIPS(squeeze) + MSE, drivers.DEFAULT_BATCH (32)
    candidates per generator forward -- the kernel shapes bench.py times -- injected eps, constant per-layer noise, 4 loop steps.  Every loss of
    the history <= 1e-3 of oracle.loss_ref.projection_literal_ref's (...sqz_MSE.py:171-184), best step exact, best latent bit-exact, and the
    seven per-tap LPIPS contributions of candidate 0 (networks_basic.py:64-92, retPerLayer) <= 1e-3 each: the whole squeeze chain at 255^2 /
    127^2 / 63^2 maps on the form-3 Winograd / pw_conv / max-pool / tap-distance kernels the 32-candidate dispatcher picks."""
    from morphganformer_amd.drivers import DEFAULT_BATCH
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.lpips import PerceptualLoss, random_squeeze_backbone
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, latent_stats, synthetic_landmarks
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    from oracle.generator_ref import generator_ref, to_torch_state
    from oracle.loss_ref import lpips_ref, mse_ref, projection_literal_ref, squeeze_backbone_random, wing_loss_ref
    cfg = FULL1024
    sd = make_state_dict(cfg, seed=0)
    G = Generator(sd, cfg, "cuda", max_batch=1)
    target = G(torch.from_numpy(synthetic_latents(cfg, 1, 1000)).cuda(), None, noise_mode="const")[0].clamp(-1, 1).clone()
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    mean, std = latent_stats(G, 10000, "cuda", gen)
    steps = 4
    lm_t, lm_s = synthetic_landmarks(steps, 1024, 7)
    eps = torch.randn(steps, 1, cfg.k, cfg.z_dim, device="cuda", generator=gen)
    P = PerceptualLoss(net="squeeze", backbone_state=random_squeeze_backbone(0))
    eng = ProjectionEngine(G, target, mean, std, ProjectionArgs(step=steps), percept=P, use_mse=True, lm_target=lm_t, lm_steps=lm_s,
                           eps=eps, noise_mode="const", use_graph=True, batch=DEFAULT_BATCH)
    lat, bstep, bloss, losses = eng.run().result()
    per_tap = P.distance_per_tap(G.img)[:, 0].cpu().numpy()          # the workspace still holds this launch sequence's candidates: row 0 = step 0
    # ---- the oracle's loop on the host cores
    tsd, bb = to_torch_state(sd), squeeze_backbone_random(0)
    lins = [l.cpu() for l in P.lins]
    tgt = target.cpu()
    taps0 = {}

    def loss_fn(i, img):
        total, layers = lpips_ref(bb, lins, img, tgt, per_layer=True)
        if i == 0:
            taps0["v"] = np.array([float(v) for v in layers])
        return float(total) + 0.01 * float(wing_loss_ref(torch.from_numpy(lm_s[i]), torch.from_numpy(lm_t))) + float(mse_ref(img, tgt))

    with torch.no_grad():
        ref = projection_literal_ref(lambda z: generator_ref(tsd, z, cfg, "const"), loss_fn, mean.cpu(), float(std), eps.cpu(), steps)
    ref_losses = np.array(ref[3])
    assert np.abs(losses - ref_losses).max() <= 1e-3 * np.abs(ref_losses).max(), (losses, ref_losses)
    assert bstep == ref[1]
    assert torch.equal(lat, ref[0])
    assert abs(bloss - ref[2]) <= 1e-3 * abs(ref[2])
    assert per_tap.shape == (7,)
    assert (np.abs(per_tap - taps0["v"]) <= 1e-3 * np.abs(taps0["v"])).all(), (per_tap, taps0["v"])


@pytest.mark.parametrize("net", ["vgg", "alex"])
def test_lpips_vgg_alex_full_size_vs_oracle(net):
    """LPIPS(vgg) -- the net 1024_example_percept_MSE.py:142-147 scores with -- and LPIPS(alex) on one 1024^2 pair: total and per-tap
    contributions against the oracle (lpips/pretrained_networks.py:58-135 topology, networks_basic.py:64-92 distance), 1e-3 each."""
    from morphganformer_amd.lpips import PerceptualLoss, random_backbone
    from oracle.loss_ref import backbone_random, lpips_ref
    torch.manual_seed(11)
    res = 1024
    x0 = (torch.rand(1, 3, res, res) * 2 - 1)
    x0 = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x0, (2, 2, 2, 2), mode="reflect"), 5, 1) * 2.5      # some spatial correlation, like an image
    x0 = x0.clamp(-1, 1)
    x1 = (x0 + 0.2 * torch.randn(1, 3, res, res)).clamp(-1, 1)
    bb_np, bb = random_backbone(net, 0), backbone_random(net, 0)
    P = PerceptualLoss(model="net-lin", net=net, use_gpu=True, backbone_state=bb_np)
    lins = [l.cpu() for l in P.lins]
    P.set_target(x1.cuda())
    out = torch.zeros(1, device="cuda")
    P.distance_into(out, x0.cuda())
    per_tap = P.distance_per_tap(x0.cuda())[:, 0].cpu().numpy()
    with torch.no_grad():
        total, layers = lpips_ref(bb, lins, x0, x1, per_layer=True, net=net)
    ref_layers = np.array([float(v) for v in layers])
    assert abs(float(out) - float(total)) <= 1e-3 * abs(float(total)), (float(out), float(total))
    assert (np.abs(per_tap - ref_layers) <= 1e-3 * np.abs(ref_layers)).all(), (per_tap, ref_layers)
    assert abs(per_tap.sum() - float(out)) <= 1e-5 * abs(float(out))


@pytest.mark.parametrize("size", [(1024, 1024), (600, 520)])
def test_lpips_merged_fire_launches_equal_the_separate_ones(size):
    """One image per forward: a Fire's expand1x1 + expand3x3 run as ONE 3x3 launch (expand1x1 in the centre tap) and their data gradients as
    one launch over the concatenated gradient (lpips.MERGE_FIRE) -- distance, per-tap contributions and the image gradient against the
    separate launches (squeezenet Fire: lpips/pretrained_networks.py:7-44)."""
    from morphganformer_amd import lpips
    torch.manual_seed(size[0])
    x0 = (torch.rand(1, 3, *size) * 2 - 1).cuda()
    x1 = (x0 + 0.3 * torch.randn(1, 3, *size).cuda()).clamp(-1, 1)
    got = {}
    keep = lpips.MERGE_FIRE
    try:
        for merge in (True, False):
            lpips.MERGE_FIRE = merge
            P = lpips.PerceptualLoss(net="squeeze", backbone_state=lpips.random_squeeze_backbone(0))
            P.set_target(x1)
            assert bool(next(iter(P._feats.values())).merge) == merge
            out, d = torch.zeros(1, device="cuda"), torch.zeros_like(x0)
            P.distance_into(out, x0, keep_taps=True)
            P.grad_into(d, scale=1.0)
            got[merge] = (float(out), P.distance_per_tap(x0)[:, 0].cpu().numpy(), d.cpu())
    finally:
        lpips.MERGE_FIRE = keep
    (va, ta, ga), (vb, tb, gb) = got[True], got[False]
    assert abs(va - vb) <= 1e-5 * abs(vb) and np.abs(ta - tb).max() <= 1e-5 * np.abs(tb).max()
    # (a ReLU input within float32 rounding of zero may take the other branch in the other arithmetic -- expand1x1 through the Winograd
    # transform instead of a plain dot product: the gate is on the bulk of the error, like the other gradient tests)
    err = (ga - gb).abs() / gb.abs().max()
    assert float(err.median()) < 1e-6 and float(err.square().mean().sqrt()) < 1e-3, (float(err.median()), float(err.square().mean().sqrt()), float(err.max()))


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
    P = PerceptualLoss(model="net-lin", net="vgg", use_gpu=True, allow_random_backbone=True)
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


@pytest.mark.parametrize("batch", [1, 5])
def test_latent_copies_variant_vs_reference_run(golden, batch):
    """projection_example_v2_percept.py:131-203 (`ProjectionArgs(latent_copies=18)`): 18 noisy copies of the latent per step, averaged like
    torch.mean before the generator; min_loss starts at 1.0 and the kept latent is the mean.  Against the run of the script's tensor
    statements on the reference generator (loop_copies_tiny.npz): every step's averaged latent BIT-exact (the kernel alone), best step
    exact, best latent bit-exact, losses 1e-3; and the kernel's summation order against torch.mean for other copy counts."""
    from morphganformer_amd import _lib
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    g = golden("loop_copies_tiny.npz")
    steps, copies = g["eps"].shape[0], int(g["copies"])
    lm = torch.from_numpy(g["latent_mean"]).reshape(17, 32)
    eng = ProjectionEngine(_tiny_gen(), torch.from_numpy(g["target"]).cuda(), lm.cuda(), float(g["latent_std"]),
                           ProjectionArgs(step=steps, beta=0.01, min_loss_init=1.0, latent_copies=copies), percept=None, use_mse=True,
                           eps=torch.from_numpy(g["eps"]).cuda(), noise_mode="const", batch=batch)
    lat, bstep, bloss, losses = eng.run().result()
    assert bstep == int(g["best_step"]) and torch.equal(lat, torch.from_numpy(g["best_latent"]))
    assert np.abs(losses - g["losses"]).max() < 1e-3 * np.abs(g["losses"]).max() and abs(bloss - float(g["best_loss"])) < 1e-3 * float(g["best_loss"])
    # the kernel on its own, all steps in one call
    out = torch.empty(steps, 17 * 32, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().mgf_latent_perturb_mean(out.data_ptr(), eng.latent_in.data_ptr(), eng.eps.data_ptr(), eng.sigma.data_ptr(), ctr.data_ptr(),
                                                  steps, steps, 17 * 32, copies, _lib.stream_ptr()), "latent_perturb_mean")
    assert np.array_equal(out.cpu().numpy().reshape(steps, 1, 17, 32), g["im_latents"])
    rng = np.random.Generator(np.random.PCG64(11))
    for n in (1, 2, 15, 16, 17, 31, 32, 33, 100, 255):
        e = torch.from_numpy(rng.standard_normal((3, n, 224)).astype(np.float32))
        base = torch.from_numpy(rng.standard_normal(224).astype(np.float32))
        sig = torch.tensor([0.5, 1.25, 0.0], dtype=torch.float32)
        want = torch.stack([torch.mean((base[None, None] + e[s][None] * float(sig[s])), 1)[0] for s in range(3)])
        got = torch.empty(3, 224, device="cuda")
        base_d, e_d, sig_d = base.cuda(), e.cuda(), sig.cuda()             # (named: a temporary's memory would be recycled by the next .cuda())
        _lib.check(_lib.lib().mgf_latent_perturb_mean(got.data_ptr(), base_d.data_ptr(), e_d.data_ptr(), sig_d.data_ptr(), ctr.data_ptr(),
                                                      3, 3, 224, n, _lib.stream_ptr()), "latent_perturb_mean")
        assert torch.equal(got.cpu(), want), n
    with pytest.raises(_lib.MgfError):
        _lib.check(_lib.lib().mgf_latent_perturb_mean(got.data_ptr(), got.data_ptr(), got.data_ptr(), got.data_ptr(), ctr.data_ptr(), 1, 1, 8, 256, _lib.stream_ptr()))


@pytest.mark.parametrize("layout", ["script", "aligned"])
def test_psnr_objective_variant(golden, layout):
    """1024_example_PSNR.py:113-114,150-158,173-175: the loss is 10 log10(255^2 / mean((p0 - p1)^2)) of the [-1, 1] images and the loop keeps
    the SMALLEST value -- every recorded loss against the script's numpy pipeline on the oracle's image (`psnr_script_ref`: the permute /
    tensor2np / flatten steps transcribed, which pair the candidate's C-H-W stream with the target's H-W-C stream; the default) or against
    the aligned definition (`psnr_layout="aligned"`, a declared deviation), best step = argmin; and a re-targeted engine follows suit."""
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.generator_ref import generator_ref, to_torch_state
    from oracle.loss_ref import psnr_ref as psnr_aligned, psnr_script_ref
    psnr_ref = (lambda img, tgt: psnr_script_ref(img, tgt)) if layout == "script" else psnr_aligned
    g = golden("loop_tiny.npz")
    steps = 7
    tsd = to_torch_state(make_state_dict(TINY, seed=0))
    assert ProjectionArgs().psnr_layout == "script"
    eng = ProjectionEngine(_tiny_gen(), torch.from_numpy(g["target"]).cuda() * 0.5, torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]),
                           ProjectionArgs(step=steps, pixel_term="psnr", psnr_layout=layout), percept=None, use_mse=True,
                           eps=torch.from_numpy(g["eps"][:steps]).cuda(), noise_mode="const", batch=3)
    eng.run()
    eng.retarget(torch.from_numpy(g["target"]).cuda(), eps=torch.from_numpy(g["eps"][:steps]).cuda())
    lat, bstep, bloss, losses = eng.run().result()
    want = []
    for i in range(steps):
        sigma = np.float32(np.float32(float(g["latent_std"])) * np.float32(0.05)) * np.float32(max(0, 1 - (i / steps) / 0.75) ** 2)
        z = torch.from_numpy(g["latent_mean"])[None] + torch.from_numpy(g["eps"][i]) * float(sigma)
        with torch.no_grad():
            img = generator_ref(tsd, z, TINY, "const")
        want.append(float(psnr_ref(img.numpy(), g["target"])))
    want = np.array(want)
    assert np.abs(losses - want).max() < 1e-3 * np.abs(want).max(), (losses, want)
    assert bstep == int(np.argmin(losses)) and bloss == float(losses.min()) and 20 < want.min() < 100
    if layout == "script":          # the two layouts are different numbers on any image that is not constant across channels and positions
        assert abs(float(psnr_aligned(g["target"] * 0.5, g["target"])) - float(psnr_script_ref(g["target"] * 0.5, g["target"]))) > 1e-3


def test_dssim_kernel_vs_oracle_and_full_size():
    """mgf_dssim_u8_f32 (`dssim`, 1024_example_SSIM.py:115-117 = lpips/__init__.py:54-55) against oracle.loss_ref.dssim_ref on the same float
    images: window-edge sizes (7x7 = one position, one past a tile, ragged), 1 and 3 channels, a shared target and one target per sample,
    scale / accumulate; identical images give exactly 0; one 1024^2 pair (the reference's size) against the oracle."""
    from morphganformer_amd import _lib
    from oracle.loss_ref import dssim_ref
    L = _lib.lib()
    rng = np.random.default_rng(11)

    def run(img, tgt, shared, scale=1.0, acc=None):
        n, c, h, w = img.shape
        out = torch.zeros(n, dtype=torch.float32, device="cuda") if acc is None else acc.clone()
        scratch = torch.empty(int(L.mgf_dssim_scratch_bytes(n, c, h, w)) // 8, dtype=torch.float64, device="cuda")
        _lib.check(L.mgf_dssim_u8_f32(out.data_ptr(), img.data_ptr(), tgt.data_ptr(), n, c, h, w, 0 if shared else c * h * w, 255.0, scale,
                                      0 if acc is None else 1, scratch.data_ptr(), _lib.stream_ptr()), "dssim")
        return out.cpu().numpy()

    for (n, c, h, w) in [(1, 1, 7, 7), (2, 3, 7, 40), (3, 3, 38, 39), (2, 1, 39, 71), (2, 3, 64, 64), (1, 3, 101, 133)]:
        base = rng.uniform(-1.2, 1.2, (c, h, w)).astype(np.float32)
        img = (base[None] + rng.normal(0, 0.2, (n, c, h, w))).astype(np.float32)
        tgts = (base[None] + rng.normal(0, 0.05, (n, c, h, w))).astype(np.float32)
        got = run(torch.from_numpy(img).cuda(), torch.from_numpy(base).cuda(), True)
        want = np.array([dssim_ref(img[i], base) for i in range(n)])
        assert np.abs(got - want).max() < 2e-7, (n, c, h, w, got, want)
        got = run(torch.from_numpy(img).cuda(), torch.from_numpy(tgts).cuda(), False, scale=0.5, acc=torch.full([n], 2.0, device="cuda"))
        want = np.array([2.0 + 0.5 * dssim_ref(img[i], tgts[i]) for i in range(n)], dtype=np.float32)
        assert np.abs(got - want).max() < 5e-7, (n, c, h, w, got, want)
        same = run(torch.from_numpy(img).cuda(), torch.from_numpy(img).cuda(), False)
        assert (same == 0).all()
    with pytest.raises(_lib.MgfError):
        run(torch.zeros(1, 3, 6, 9, device="cuda"), torch.zeros(3, 6, 9, device="cuda"), True)
    # the reference's size: smooth images with noise (the statistics of a face, not of white noise)
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32) / 1024
    base = np.stack([np.sin(6 * xx + 3 * yy), np.cos(5 * yy) * xx, xx * yy * 2 - 1]).astype(np.float32)
    img = (base[None] + rng.normal(0, 0.08, (2, 3, 1024, 1024))).astype(np.float32)
    got = run(torch.from_numpy(img).cuda(), torch.from_numpy(base).cuda(), True)
    want = np.array([dssim_ref(img[i], base) for i in range(2)])
    assert np.abs(got - want).max() < 2e-7 and 0.05 < want.min() < 0.5, (got, want)


def test_dssim_objective_variant(golden):
    """1024_example_SSIM.py:115-117,173-175: the loop keeps the candidate with the smallest (1 - SSIM) / 2 of the uint8 images -- every
    recorded loss against the oracle's dssim on the oracle's image (a pixel within 1e-3 of a rounding boundary may quantise one level apart:
    1e-3 on a loss of order 0.3), best step = argmin."""
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.generator_ref import generator_ref, to_torch_state
    from oracle.loss_ref import dssim_ref
    g = golden("loop_tiny.npz")
    steps = 7
    tsd = to_torch_state(make_state_dict(TINY, seed=0))
    eng = ProjectionEngine(_tiny_gen(), torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]),
                           ProjectionArgs(step=steps, pixel_term="dssim"), percept=None, use_mse=True, eps=torch.from_numpy(g["eps"][:steps]).cuda(),
                           noise_mode="const", batch=3)
    lat, bstep, bloss, losses = eng.run().result()
    want = []
    for i in range(steps):
        sigma = np.float32(np.float32(float(g["latent_std"])) * np.float32(0.05)) * np.float32(max(0, 1 - (i / steps) / 0.75) ** 2)
        z = torch.from_numpy(g["latent_mean"])[None] + torch.from_numpy(g["eps"][i]) * float(sigma)
        with torch.no_grad():
            img = generator_ref(tsd, z, TINY, "const")
        want.append(float(dssim_ref(img.numpy()[0], g["target"].reshape(img.shape[1:]))))
    want = np.array(want)
    assert np.abs(losses - want).max() < 1e-3, (losses, want)
    assert bstep == int(np.argmin(losses)) and bloss == float(losses.min()) and 0 < want.min() < 0.5


def test_lbp_kernels_bit_exact_vs_oracle():
    """csrc/lbp.hip against oracle.loss_ref step by step and bit for bit: the 224 x 224 gray image (to_pil quantisation, BGR2GRAY weights in
    both channel orders, fixed-point resize) from an odd-sized and from a 1024^2 image, the LBP(24, 3, 'uniform') code map -- smooth,
    noisy and SATURATED (flat) regions, where skimage's interpolation ties at the ulp level decide the code --, and the float64 cosine
    distance; the target feature of a file's pixels (lbp.target_feature)."""
    from morphganformer_amd import lbp
    from oracle.loss_ref import (cv_bgr2gray_u8_ref, cv_resize_linear_gray_ref, lbp_cosine_distance_ref, lbp_feature_file_ref,
                                 lbp_feature_im_ref, lbp_uniform_ref, to_u8_ref)
    rng = np.random.default_rng(21)
    for (n, h, w) in ((3, 300, 280), (1, 1024, 1024), (2, 64, 64)):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.stack([np.stack([1.4 * np.sin(xx / (9 + 3 * k + c)) * np.cos(yy / (13 + c)) for c in range(3)]) for k in range(n)]).astype(np.float32)
        img += rng.normal(0, 0.05, img.shape).astype(np.float32)              # |1.4 sin| > 1 in places: clipped to 0 / 255, i.e. flat patches
        ws = lbp.LbpWorkspace(n, h, w, "cuda")
        x = torch.from_numpy(img).cuda()
        for true_order in (False, True):
            g = ws.gray224(x, true_rgb_order=true_order).cpu().numpy().reshape(n, 224, 224)
            for i in range(n):
                want = cv_resize_linear_gray_ref(cv_bgr2gray_u8_ref(to_u8_ref(img[i]).transpose(1, 2, 0), blue_first=not true_order), 224, 224)
                assert np.array_equal(g[i], want), (n, h, w, true_order, i)
        gray = ws.gray224(x)
        codes = ws.codes(gray).cpu().numpy().reshape(n, 224, 224)
        feats = [lbp_feature_im_ref(img[i]) for i in range(n)]
        for i in range(n):
            assert np.array_equal(codes[i].reshape(-1).astype(np.float64), feats[i]), (n, h, w, i, int((codes[i].reshape(-1) != feats[i]).sum()))
        assert len(np.unique(codes)) > 20
        tgt = torch.from_numpy(codes[0].reshape(-1)).cuda()
        out = torch.zeros(n, dtype=torch.float64, device="cuda")
        ws.distance_into(out, x, tgt)
        want = np.array([lbp_cosine_distance_ref(feats[i], feats[0]) for i in range(n)])
        assert np.array_equal(out.cpu().numpy(), want), (out.cpu().numpy(), want)
        assert n == 1 or want[1] > 1e-3
    # argument checks happen on the host, before any launch
    from morphganformer_amd import _lib
    Lh = _lib.lib()
    assert Lh.mgf_lbp_scratch_bytes(0) == 0 and Lh.mgf_lbp_scratch_bytes(3) == 3 * 196 * 16
    assert Lh.mgf_lbp_codes_u8(None, gray.data_ptr(), ws.off.data_ptr(), 1, None) != 0 and b"lbp_codes" in Lh.mgf_last_error()
    assert Lh.mgf_lbp_distance_f64(out.data_ptr(), gray.data_ptr(), tgt.data_ptr(), ws.off.data_ptr(), 0, ws.scratch.data_ptr(), None) != 0
    # the target side: a file's own pixels, true colour order, any size
    u8 = rng.integers(0, 256, (150, 131, 3)).astype(np.uint8)
    u8[20:60, 30:90] = 255
    got = lbp.target_feature(u8).cpu().numpy().astype(np.float64)
    assert np.array_equal(got, lbp_feature_file_ref(u8))


def test_lbp_objective_variant(golden):
    """1024_example_LBP_percept.py:151-172: the loop keeps the candidate with the smallest float64 LBP matching distance to the target
    file's feature.  One batch of candidates: every recorded loss EQUALS the oracle's distance on the image the engine generated (the
    same latents through the same generator call), differs from the oracle's on the oracle generator's image only by what a pixel near
    a rounding boundary can move, and the best step is the argmin; a re-targeted engine scores the new feature."""
    from morphganformer_amd import lbp
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.generator_ref import generator_ref, to_torch_state
    from oracle.loss_ref import lbp_cosine_distance_ref, lbp_feature_file_ref, lbp_feature_im_ref, to_u8_ref
    from morphganformer_amd import _lib
    g = golden("loop_tiny.npz")
    steps = 4
    tsd = to_torch_state(make_state_dict(TINY, seed=0))
    file_px = to_u8_ref(g["target"][0]).transpose(1, 2, 0)                   # "the target file": the golden target as saved pixels
    feat_t = lbp_feature_file_ref(file_px)
    G = _tiny_gen()
    args = ProjectionArgs(step=steps, pixel_term="lbp", min_loss_init=1.0)
    mk = lambda **kw: ProjectionEngine(G, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]), args,
                                       percept=None, eps=torch.from_numpy(g["eps"][:steps]).cuda(), noise_mode="const", batch=steps, **kw)
    with pytest.raises(_lib.MgfError):
        mk()                                                                 # the feature of the target file is the objective: it must be given
    eng = mk(lbp_target=lbp.target_feature(file_px))
    lat, bstep, bloss, losses = eng.run().result()
    sig = [np.float32(np.float32(float(g["latent_std"])) * np.float32(0.05)) * np.float32(max(0, 1 - (i / steps) / 0.75) ** 2) for i in range(steps)]
    z = torch.cat([torch.from_numpy(g["latent_mean"])[None] + torch.from_numpy(g["eps"][i]) * float(sig[i]) for i in range(steps)]).cuda()
    own = G.forward_workspace(z, args.truncation_psi, noise_mode="const", lean=True)[0].cpu().numpy()
    want_own = np.array([lbp_cosine_distance_ref(lbp_feature_im_ref(own[i]), feat_t) for i in range(steps)])
    assert np.array_equal(losses, want_own), (losses, want_own)
    with torch.no_grad():
        ref_img = generator_ref(tsd, z.cpu(), TINY, "const").numpy()
    want_ref = np.array([lbp_cosine_distance_ref(lbp_feature_im_ref(ref_img[i]), feat_t) for i in range(steps)])
    assert np.abs(losses - want_ref).max() < 2e-2 and 0 < want_ref.min() < 1, (losses, want_ref)
    assert bstep == int(np.argmin(losses)) and bloss == float(losses.min())
    # re-target: another file -> another feature, same launch sequence
    other = np.ascontiguousarray(file_px[::-1])
    eng.retarget(torch.from_numpy(g["target"]).cuda(), eps=torch.from_numpy(g["eps"][:steps]).cuda(), lbp_target=lbp.target_feature(other))
    l2 = eng.run().result()[3]
    f2 = lbp_feature_file_ref(other)
    assert np.array_equal(l2, np.array([lbp_cosine_distance_ref(lbp_feature_im_ref(own[i]), f2) for i in range(steps)]))


def test_v1_pooled_percept_objective(golden):
    """projection_example_v1.py:148-160: a generated image taller than 256 px is block-averaged by height // 256 before LPIPS, against a
    target given at the pooled size -- here the 64^2 generator with pool_above=32 (factor 2), LPIPS(squeeze) only: every recorded loss against
    the oracle's pooling + LPIPS on the oracle's image; a target at the wrong size is refused."""
    from morphganformer_amd import _lib
    from morphganformer_amd.lpips import PerceptualLoss, random_squeeze_backbone
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.generator_ref import generator_ref, to_torch_state
    from oracle.loss_ref import lpips_ref, pool_above_ref, squeeze_backbone_random
    g = golden("loop_tiny.npz")
    steps = 5
    tsd = to_torch_state(make_state_dict(TINY, seed=0))
    target = torch.from_numpy(pool_above_ref(g["target"], 32).astype(np.float32))
    assert tuple(target.shape) == (1, 3, 32, 32)
    P = PerceptualLoss(net="squeeze", backbone_state=random_squeeze_backbone(0))
    args = ProjectionArgs(step=steps, pool_above=32)
    mk = lambda tgt: ProjectionEngine(_tiny_gen(), tgt.cuda(), torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]), args, percept=P,
                                      use_mse=False, eps=torch.from_numpy(g["eps"][:steps]).cuda(), noise_mode="const", batch=2)
    lat, bstep, bloss, losses = mk(target).run().result()
    bb = squeeze_backbone_random(0)
    lins = [l.cpu() for l in P.lins]
    for i in range(steps):
        sigma = np.float32(np.float32(float(g["latent_std"])) * np.float32(0.05)) * np.float32(max(0, 1 - (i / steps) / 0.75) ** 2)
        z = torch.from_numpy(g["latent_mean"])[None] + torch.from_numpy(g["eps"][i]) * float(sigma)
        with torch.no_grad():
            img = torch.from_numpy(pool_above_ref(generator_ref(tsd, z, TINY, "const").numpy(), 32))
            want = float(lpips_ref(bb, lins, img, target).sum())
        assert abs(losses[i] - want) < 1e-3 * abs(want), (i, losses[i], want)
    assert bstep == int(np.argmin(losses))
    with pytest.raises(_lib.MgfError, match="image-space losses see 32x32"):
        mk(torch.from_numpy(g["target"]))


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


LPIPS_CHNS = {"squeeze": [64, 128, 256, 384, 384, 512, 512], "vgg": [64, 128, 256, 512, 512], "alex": [64, 192, 384, 256, 256]}


def test_lpips_distance_half_vs_reference_fixture(golden):
    """tests/golden/lpips_dist.npz = the REFERENCE's own lpips code (PNetLin.forward, ScalingLayer, NetLinLayer with the vendored weights,
    normalize_tensor, spatial_average: networks_basic.py:64-111, lpips/__init__.py:26-46) run on injected tap tensors.  The HIP
    distance kernels (mgf_lpips_unit_f32 / mgf_lpips_layer_f32, the lin heads shipped in weights/) against those outputs, for all
    three nets; then the whole PerceptualLoss.forward (HIP SqueezeNet backbone on seeded weights + HIP distance) against the
    reference forward with the same backbone injected."""
    from morphganformer_amd import _lib
    from morphganformer_amd.lpips import PerceptualLoss, random_squeeze_backbone
    g = golden("lpips_dist.npz")
    L, st = _lib.lib(), _lib.stream_ptr()
    scratch = torch.empty(2 * int(L.mgf_reduce_scratch_floats()), device="cuda")
    for net, chns in LPIPS_CHNS.items():
        P = PerceptualLoss(model="net-lin", net=net, use_gpu=True, allow_random_backbone=True)
        ref_lin = golden(f"lpips_lin_{net}.npz")
        out = torch.zeros(2, device="cuda")
        per_layer = []
        for i, c in enumerate(chns):
            assert np.array_equal(P.lins[i].cpu().numpy(), ref_lin[f"lin{i}"])          # shipped heads == the reference's vendored data
            a, b = torch.from_numpy(g[f"{net}_tap0_{i}"]).cuda(), torch.from_numpy(g[f"{net}_tap1_{i}"]).cuda()
            n, _, h, w = a.shape
            bu = torch.empty_like(b)
            _lib.check(L.mgf_lpips_unit_f32(bu.data_ptr(), b.data_ptr(), n, c, h * w, st), "lpips_unit")
            if i == 1:
                au = torch.empty_like(a)
                _lib.check(L.mgf_lpips_unit_f32(au.data_ptr(), a.data_ptr(), n, c, h * w, st), "lpips_unit")
                assert float((au.cpu() - torch.from_numpy(g[f"{net}_unit0_1"])).abs().max()) < 2e-6
                assert float(au[:, :, 0, 0].abs().max()) == 0.0                          # all-zero pixel: 0 / (0 + 1e-10)
            one = torch.zeros(2, device="cuda")
            _lib.check(L.mgf_lpips_layer_f32(one.data_ptr(), a.data_ptr(), bu.data_ptr(), P.lins[i].data_ptr(), n, c, h * w, c * h * w, 0,
                                             scratch.data_ptr(), st), "lpips_layer")
            _lib.check(L.mgf_lpips_layer_f32(out.data_ptr(), a.data_ptr(), bu.data_ptr(), P.lins[i].data_ptr(), n, c, h * w, c * h * w, 1,
                                             scratch.data_ptr(), st), "lpips_layer")
            per_layer.append(one.cpu().numpy())
            if i > 0:        # (the reference's res[0] is aliased to the running total: networks_basic.py:85-87)
                want = g[f"{net}_res_{i}"].reshape(-1)
                assert np.abs(per_layer[-1] - want).max() < 1e-5 * np.abs(want).max(), (net, i)
        want = g[f"{net}_val"].reshape(-1)
        assert np.abs(out.cpu().numpy() - want).max() < 1e-5 * np.abs(want).max(), net
    P = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, backbone_state=random_squeeze_backbone(0))
    a, b = torch.from_numpy(g["full_in0"]).cuda(), torch.from_numpy(g["full_in1"]).cuda()
    got = P(a, b).cpu().numpy()
    assert got.shape == g["full_val"].shape and np.abs(got - g["full_val"]).max() < 1e-3 * np.abs(g["full_val"]).max()
    got = P((a + 1) / 2, (b + 1) / 2, normalize=True).cpu().numpy()
    assert np.abs(got - g["full_val_normalize"]).max() < 1e-3 * np.abs(g["full_val_normalize"]).max()


def _config3_objective(G, target, cfg, steps, batch, lm, eps, mean, std, depth=18, gamma=1e-4, noise_mode="const", seed=0):
    """BASELINE config 3's objective: Wing + IResNet embedding MSE + LPIPS(squeeze) + MSE."""
    from morphganformer_amd.iresnet import BiometricLoss, IResNetEmbedder, random_state
    from morphganformer_amd.lpips import PerceptualLoss, random_squeeze_backbone
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    bio_sd = random_state(depth, 1)
    P = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, backbone_state=random_squeeze_backbone(0))
    bio = BiometricLoss(IResNetEmbedder(bio_sd, depth=depth, n=batch))
    eng = ProjectionEngine(G, target, mean, std, ProjectionArgs(step=steps, min_loss_init=1e30), percept=P, use_mse=True, lm_target=lm[0],
                           lm_steps=lm[1], eps=eps, noise_mode=noise_mode, batch=batch, biometric=bio, gamma=gamma, seed=seed)
    return eng, bio_sd


@pytest.mark.parametrize("batch", [1, 3])
def test_config3_four_term_objective_vs_oracle(golden, batch):
    """Config 3 of BASELINE.json (Wing + biometric + LPIPS + MSE) on the tiny generator against the CPU oracle's literal loop with the
    same four terms: best step exact, best latent bit-exact, every loss of the history to 1e-3 (sequential and batched steps)."""
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.embed_ref import biometric_loss_ref
    from oracle.generator_ref import generator_ref, to_torch_state
    from oracle.loss_ref import lpips_ref, mse_ref, projection_literal_ref, squeeze_backbone_random, wing_loss_ref
    g = golden("loop_tiny.npz")
    steps, gamma = 8, 1e-4
    sd = make_state_dict(TINY, seed=0)
    tsd = to_torch_state(sd)
    G = _tiny_gen()
    target = torch.from_numpy(g["target"])
    mean, std = torch.from_numpy(g["latent_mean"]), float(g["latent_std"])
    eps = torch.from_numpy(g["eps"][:steps])
    lm = (g["lm_target"], g["lm_steps"][:steps])
    eng, bio_sd = _config3_objective(G, target.cuda(), TINY, steps, batch, lm, eps.cuda(), mean.cuda(), std, gamma=gamma)
    lat, bstep, bloss, losses = eng.run().result()
    bb = squeeze_backbone_random(0)
    lin = golden("lpips_lin_squeeze.npz")
    lins = [torch.from_numpy(lin[f"lin{i}"]) for i in range(7)]
    bsd = {k: torch.from_numpy(v) for k, v in bio_sd.items()}

    def loss_fn(i, img):
        with torch.no_grad():
            p = float(lpips_ref(bb, lins, img, target).sum())
            w = float(wing_loss_ref(torch.from_numpy(lm[1][i]), torch.from_numpy(lm[0])))
            b = float(biometric_loss_ref(bsd, img, target, 18))
            return p + gamma * b + 0.01 * w + 1.0 * float(mse_ref(img, target))

    with torch.no_grad():
        ref = projection_literal_ref(lambda z: generator_ref(tsd, z, TINY, "const"), loss_fn, mean, std, eps, steps, min_loss_init=1e30)
    want = np.array(ref[3], np.float64)
    assert np.abs(losses - want).max() < 1e-3 * np.abs(want).max(), (losses, want)
    assert bstep == ref[1]
    assert torch.equal(lat, ref[0]), "best latent must be bit-exact under injected noise"
    assert abs(bloss - ref[2]) < 1e-3 * abs(ref[2])


def test_config3_full_size_properties():
    """Config 3's four-term objective at the full 1024^2 size, IResNet-50 embedder: size-independent properties of a 12-step run --
    best loss == min of the history, best step == argmin, reproducible history, the selected latent is mean + eps*sigma, and the
    biometric term really is in the total (history minus the three-term history is positive and equals gamma * the embedding MSE)."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.lpips import PerceptualLoss, random_squeeze_backbone
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, latent_stats, synthetic_landmarks
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    G = Generator(make_state_dict(FULL1024, seed=0), FULL1024, "cuda", max_batch=1)
    target = G(torch.from_numpy(synthetic_latents(FULL1024, 1, 1000)).cuda(), None, noise_mode="const")[0].clamp(-1, 1).clone()
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    mean, std = latent_stats(G, 10000, "cuda", gen)
    steps, batch, gamma = 12, 4, 1e-6
    lm = synthetic_landmarks(steps, 1024, 7)
    gen.manual_seed(5)
    eps = torch.randn(steps, 1, FULL1024.k, FULL1024.z_dim, device="cuda", generator=gen)
    hist = []
    for _ in range(2):
        eng, _sd = _config3_objective(G, target, FULL1024, steps, batch, lm, eps, mean, std, depth=50, gamma=gamma)
        lat, bstep, bloss, losses = eng.run().result()
        hist.append(losses)
        assert not np.isnan(losses).any()
        assert bstep == int(np.argmin(losses)) and bloss == float(losses.min())
    assert np.array_equal(hist[0], hist[1])
    assert torch.equal(lat.cuda(), eng.latent_in + eng.eps[bstep] * eng.sigma[bstep])
    P = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, backbone_state=random_squeeze_backbone(0))
    three = ProjectionEngine(G, target, mean, std, ProjectionArgs(step=steps, min_loss_init=1e30), percept=P, lm_target=lm[0], lm_steps=lm[1],
                             eps=eps, noise_mode="const", batch=batch).run().result()[3]
    extra = hist[0] - three
    assert (extra > 0).all()
    img = G((eng.latent_in + eng.eps[bstep] * eng.sigma[bstep]), None, noise_mode="const")[0]
    want = gamma * float(eng.biometric(img[:1], target))
    assert abs(extra[bstep] - want) < 1e-3 * want + 1e-6 * abs(hist[0][bstep]), (extra[bstep], want)
