"""GPU parity of the biometric branch: IResNet embedder + embedding-MSE loss vs the reference module's
output (tests/golden/iresnet18.npz) and the CPU oracle; the three small ops of csrc/embed.hip vs torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_embed_small_ops_vs_torch():
    from morphganformer_amd import _lib
    L, st = _lib.lib(), _lib.stream_ptr()
    torch.manual_seed(0)
    x = torch.randn(3, 5, 7, 9)
    sc, sh, sl = torch.rand(5) + 0.5, torch.randn(5), torch.rand(5) * 0.4
    y = torch.empty_like(x).cuda()
    xd, scd, shd, sld = x.cuda(), sc.cuda(), sh.cuda(), sl.cuda()          # keep the device copies alive across the launches
    _lib.check(L.mgf_channel_affine_prelu_f32(y.data_ptr(), xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), sld.data_ptr(), 3, 5, 63, st))
    want = torch.nn.functional.prelu(x * sc.reshape(1, 5, 1, 1) + sh.reshape(1, 5, 1, 1), sl)
    assert torch.equal(y.cpu(), want)
    _lib.check(L.mgf_channel_affine_prelu_f32(y.data_ptr(), xd.data_ptr(), None, None, sld.data_ptr(), 3, 5, 63, st))
    assert torch.equal(y.cpu(), torch.nn.functional.prelu(x, sl))
    # linear
    xi, w, b = torch.randn(5, 1000), torch.randn(37, 1000) / 30, torch.randn(37)
    out = torch.empty(5, 37).cuda()
    xid, wd, bd = xi.cuda(), w.cuda(), b.cuda()
    _lib.check(L.mgf_linear_f32(out.data_ptr(), xid.data_ptr(), wd.data_ptr(), bd.data_ptr(), 5, 1000, 37, st))
    ref = torch.nn.functional.linear(xi.double(), w.double(), b.double())
    assert float((out.cpu().double() - ref).abs().max()) < 1e-5
    assert L.mgf_linear_f32(out.data_ptr(), xid.data_ptr(), wd.data_ptr(), None, 17, 1000, 37, st) != 0     # > 16 rows
    # bilinear resize, align_corners=False (down and up)
    for (ih, iw, oh, ow) in ((1024, 1024, 112, 112), (64, 48, 112, 112), (131, 77, 50, 201)):
        img = torch.randn(2, 3, ih, iw)
        o = torch.empty(2, 3, oh, ow).cuda()
        imgd = img.cuda()
        _lib.check(L.mgf_resize_bilinear_f32(o.data_ptr(), imgd.data_ptr(), 6, ih, iw, oh, ow, st))
        want = torch.nn.functional.interpolate(img, size=(oh, ow), mode="bilinear", align_corners=False)
        assert float((o.cpu() - want).abs().max()) < 2e-5, (ih, iw, oh, ow)


def test_iresnet18_matches_reference_module_output(golden):
    from morphganformer_amd.iresnet import IResNetEmbedder, random_state
    g = golden("iresnet18.npz")
    net = IResNetEmbedder(random_state(18, 0), depth=18, n=2)
    emb = net.embed(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert emb.shape == (2, 512)
    assert np.abs(emb - g["embedding"]).max() < 2e-4 * np.abs(g["embedding"]).max()
    # batch size change re-allocates; same result row by row
    e1 = net.embed(torch.from_numpy(g["x"][1:]).cuda()).cpu().numpy()
    assert np.abs(e1[0] - emb[1]).max() < 1e-5 * np.abs(emb).max()


@pytest.mark.parametrize("depth", [18, 50])
def test_biometric_loss_vs_oracle_and_in_the_loop(golden, depth):
    from morphganformer_amd.iresnet import BiometricLoss, IResNetEmbedder, random_state
    from oracle.embed_ref import biometric_loss_ref
    torch.manual_seed(depth)
    pred = (torch.rand(3, 3, 160, 160) * 2 - 1)
    tgt = (pred[:1] + 0.2 * torch.randn(1, 3, 160, 160)).clamp(-1, 1)
    sd_np = random_state(depth, 1)
    B = BiometricLoss(IResNetEmbedder(sd_np, depth=depth, n=3))
    got = B(pred.cuda(), tgt.cuda()).cpu()
    with torch.no_grad():
        want = biometric_loss_ref({k: torch.from_numpy(v) for k, v in sd_np.items()}, pred, tgt, depth)
    assert float((got - want).abs().max()) < 1e-3 * float(want.abs().max()), (got, want)
    assert float(B(tgt.cuda(), tgt.cuda())) == 0.0
    if depth != 18:
        return
    # inside the projection loop: total = beta*MSE + gamma*embedding-MSE, best-of selection as usual
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    g = golden("loop_tiny.npz")
    G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
    steps, gamma = 6, 1e-4
    mk = lambda bio: ProjectionEngine(G, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(),
                                      float(g["latent_std"]), ProjectionArgs(step=steps, min_loss_init=1e30), eps=torch.from_numpy(g["eps"][:steps]).cuda(),
                                      noise_mode="const", batch=3, biometric=bio, gamma=gamma)
    base = mk(None).run().result()[3]
    with_bio = mk(BiometricLoss(IResNetEmbedder(sd_np, depth=18, n=3))).run().result()[3]
    extra = with_bio - base
    assert (extra > 0).all()
    lat = torch.from_numpy(g["latent_mean"])[None] + torch.from_numpy(g["eps"][0]) * float(
        np.float32(float(g["latent_std"]) * 0.05 * max(0, 1 - 0 / 0.75) ** 2))
    img0 = G(lat.cuda(), None, noise_mode="const")[0].cpu()
    with torch.no_grad():
        want0 = gamma * float(biometric_loss_ref({k: torch.from_numpy(v) for k, v in sd_np.items()}, img0, torch.from_numpy(g["target"]), 18))
    assert abs(extra[0] - want0) < 2e-3 * want0


def test_facenet_head_ops_vs_torch():
    """mgf_spatial_mean_f32 / mgf_l2_normalize_f32 (AdaptiveAvgPool2d(1), F.normalize) and the 1x1 GEMM's ReLU-behind-the-residual
    epilogue (MGF_ACT_RELU_POST: the closing conv of an InceptionResnetV1 residual block) against torch in float64."""
    from morphganformer_amd import _lib, conv as cv
    L, st = _lib.lib(), _lib.stream_ptr()
    torch.manual_seed(1)
    x = torch.randn(3, 7, 30, 29)
    xd, m = x.cuda(), torch.empty(21).cuda()
    _lib.check(L.mgf_spatial_mean_f32(m.data_ptr(), xd.data_ptr(), 21, 30 * 29, st))
    assert float((m.cpu().double() - x.double().mean((2, 3)).reshape(-1)).abs().max()) < 1e-6
    v = torch.randn(5, 512)
    v[3] = 0                                                              # a zero row: eps keeps it zero, like F.normalize
    vd, o = v.cuda(), torch.empty(5, 512).cuda()
    _lib.check(L.mgf_l2_normalize_f32(o.data_ptr(), vd.data_ptr(), 5, 512, 1e-12, st))
    assert float((o.cpu().double() - torch.nn.functional.normalize(v.double(), p=2, dim=1)).abs().max()) < 1e-6
    for (cin, cout, hw) in ((96, 256, (25, 25)), (256, 896, (14, 14)), (384, 1792, (6, 7))):
        xx, w, b, r = torch.randn(2, cin, *hw), torch.randn(cout, cin, 1, 1) / cin ** 0.5, torch.randn(cout), torch.randn(2, cout, *hw)
        pc = cv.pack_weights(w.cuda())
        rd, bd = r.cuda(), b.cuda()
        y = cv.conv_forward(xx.cuda(), pc, epilogue=_lib.make_epilogue(bias=bd, act="relu_post", residual=rd))
        want = torch.relu(torch.nn.functional.conv2d(xx.double(), w.double(), b.double()) + r.double())
        assert float((y.cpu().double() - want).abs().max()) < 2e-5 * float(want.abs().max()), (cin, cout)
        assert float(y.min()) == 0.0
    with pytest.raises(_lib.MgfError, match="RELU_POST"):
        cv.conv_forward(xx.cuda(), pc, epilogue=_lib.make_epilogue(bias=bd, act="relu_post"))          # no residual: refused


@pytest.mark.parametrize("n,size", [(2, 160), (3, (99, 131)), (1, 1024)])
def test_facenet_inception_resnet_v1_vs_oracle(n, size):
    """facenet.InceptionResnetV1Embedder -- the network 1024_example_FaceNet_percept.py:30-32 scores with, fed the UN-RESIZED image
    (:147-158) -- against the oracle's restatement of the published facenet_pytorch topology on seeded weights: embedding <= 1e-3, the
    stage outputs <= 1e-4 (160^2: all Winograd / tap-list / 1x1 / pool launches; an odd non-square size; the driver's 1024^2).
    Parity with the real package is UNPINNED (absent offline)."""
    from morphganformer_amd.facenet import InceptionResnetV1Embedder, random_state, conv_gflop
    from oracle.embed_ref import inception_resnet_v1_ref
    h, w = (size, size) if isinstance(size, int) else size
    torch.manual_seed(h + w)
    x = (torch.rand(n, 3, h, w) * 2 - 1)
    x = (torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1) * 1.7).clamp(-1, 1)
    sd_np = random_state(3)
    net = InceptionResnetV1Embedder(sd_np, n=n)
    emb = net(x.cuda()).cpu()
    taps = {}
    with torch.no_grad():
        want = inception_resnet_v1_ref({k: torch.from_numpy(v) for k, v in sd_np.items()}, x, taps)
    assert tuple(emb.shape) == (n, 512)
    assert float((emb.norm(dim=1) - 1).abs().max()) < 1e-5
    got_taps = {"stem": net.bufs["stem"][-1], "repeat_1": net.bufs["stages"][0]["blocks"][-1]["x"], "repeat_2": net.bufs["stages"][2]["blocks"][-1]["x"],
                "block8": net.bufs["stages"][4]["blocks"][-1]["x"]}
    for k, t in got_taps.items():
        assert tuple(t.shape) == tuple(taps[k].shape), k
        assert float((t.cpu() - taps[k]).abs().max()) < 1e-4 * float(taps[k].abs().max()), k
    assert float((emb - want).abs().max()) < 1e-3 * float(want.abs().max())
    if h == 1024:
        assert 160 < conv_gflop(h, w) < 170                        # 164.6 GFLOP per 1024^2 image: what config 3's biometric term costs
    # a second call with another batch size re-allocates and gives the same rows
    if n > 1:
        e1 = net(x[1:2].cuda()).cpu()
        assert float((e1[0] - emb[1]).abs().max()) < 1e-5


def test_facenet_biometric_term_in_the_loop(golden):
    """BiometricLoss(embedder="facenet") inside the literal loop on the 256^2 generator (config 3's four-term objective with the embedder
    the driver calls): every recorded loss against beta * MSE + gamma * facenet_loss_ref of the oracle, best step = argmin."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.facenet import random_state
    from morphganformer_amd.iresnet import BiometricLoss
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import SMALL256, make_state_dict
    from oracle.embed_ref import facenet_loss_ref
    from oracle.loss_ref import mse_ref
    g = golden("loop_config0_256.npz")
    G = Generator(make_state_dict(SMALL256, seed=0), SMALL256, "cuda", max_batch=1)
    target = torch.from_numpy(g["target_u8"]).float().div(255).sub(0.5).div(0.5)[None].cuda()
    steps, gamma, batch = 5, 10.0, 3
    sd_np = random_state(5)
    bio = BiometricLoss("facenet", state=sd_np, n=batch)
    eng = ProjectionEngine(G, target, torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]),
                           ProjectionArgs(step=steps, min_loss_init=1e30), eps=torch.from_numpy(g["eps"][:steps]).cuda(),
                           noise_mode="const", batch=batch, biometric=bio, gamma=gamma)
    lat, bstep, bloss, losses = eng.run().result()
    tsd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    for i in range(steps):
        sigma = np.float32(np.float32(float(g["latent_std"])) * np.float32(0.05)) * np.float32(max(0, 1 - (i / steps) / 0.75) ** 2)
        z = torch.from_numpy(g["latent_mean"])[None] + torch.from_numpy(g["eps"][i]) * float(sigma)
        img = G(z.cuda(), None, noise_mode="const")[0].cpu()
        with torch.no_grad():
            want = float(mse_ref(img, target.cpu())) + gamma * float(facenet_loss_ref(tsd, img, target.cpu()))
        assert abs(losses[i] - want) < 1e-3 * abs(want), (i, losses[i], want)
    assert bstep == int(np.argmin(losses))


@pytest.mark.parametrize("n,size", [(2, 160), (1, (99, 131))])
def test_facenet_gradient_matches_autograd(n, size):
    """d(scale * MSE(embed(pred), embed(target)))/d(pred) through the InceptionResnetV1 embedder vs FLOAT64 autograd through the oracle's
    restatement.  ~130 ReLU layers make the gradient piecewise: a pre-activation within float32 rounding of zero takes the other branch in
    any float32 implementation, so the gate is on the bulk of the error distribution (median, rms), like the IResNet test."""
    from morphganformer_amd.facenet import random_state
    from morphganformer_amd.iresnet import BiometricLoss
    from oracle.embed_ref import facenet_loss_ref
    h, w = (size, size) if isinstance(size, int) else size
    torch.manual_seed(h * w + n)
    sd_np = random_state(9)
    pred = (torch.rand(n, 3, h, w, dtype=torch.float64) * 2 - 1).requires_grad_(True)
    target = torch.rand(1, 3, h, w, dtype=torch.float64) * 2 - 1
    val = facenet_loss_ref({k: torch.from_numpy(v).double() for k, v in sd_np.items()}, pred, target.expand(n, -1, -1, -1))
    (ref,) = torch.autograd.grad(val.sum() * 3e6, pred)          # (unit-norm embeddings of random nets: the raw gradient is ~1e-7)
    bio = BiometricLoss("facenet", state=sd_np, n=n)
    bio.embedder.keep_activations = True
    bio.set_target(target.float().cuda())
    out = torch.empty(n, device="cuda")
    bio.distance_into(out, pred.detach().float().cuda())
    assert float((out.cpu().double() - val.detach()).abs().max()) < 1e-3 * float(val.detach().abs().max())
    dimg = torch.full((n, 3, h, w), 0.5, device="cuda")
    bio.grad_into(dimg, scale=3e6, accumulate=True)
    first = dimg.clone()
    bio.grad_into(dimg, scale=3e6)
    assert torch.allclose(first - 0.5, dimg, rtol=0, atol=1e-5 * float(ref.abs().max()))      # accumulate adds to what was there
    err = (dimg.double().cpu() - ref).abs() / ref.abs().max()
    assert float(err.median()) < 2e-5 and float(err.square().mean().sqrt()) < 2e-3, (float(err.median()), float(err.square().mean().sqrt()))
    # without keep_activations the buffers were re-used block after block: the backward refuses
    plain = BiometricLoss("facenet", state=sd_np, n=n)
    plain.set_target(target.float().cuda())
    plain.distance_into(out, pred.detach().float().cuda())
    from morphganformer_amd import _lib
    with pytest.raises(_lib.MgfError, match="keep_activations"):
        plain.grad_into(dimg)


def test_gradient_projection_with_the_facenet_term():
    """Gradient mode with BASELINE config 3's embedder: GradientProjectionEngine(biometric=BiometricLoss("facenet")) on the 256^2 generator
    runs as a replayed graph, its first loss equals the literal engine's loss of the same candidate, and Adam moves the latent."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.iresnet import BiometricLoss
    from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import SMALL256, make_state_dict
    cfg = SMALL256
    G = Generator(make_state_dict(cfg, seed=0), cfg, "cuda", max_batch=1)
    steps = 5
    torch.manual_seed(4)
    latent_mean = torch.randn(cfg.k, cfg.z_dim, device="cuda")
    eps = torch.randn(steps, 1, cfg.k, cfg.z_dim, device="cuda")
    target = G(torch.randn(1, cfg.k, cfg.z_dim, device="cuda"), None, noise_mode="const")[0].clamp(-1, 1).clone()
    args = ProjectionArgs(step=steps, lr=0.05, lr_rampup=0.2, min_loss_init=1e30)
    mk = lambda cls, **kw: cls(G, target, latent_mean, 1.0, args, percept=None, eps=eps, noise_mode="const",
                               biometric=BiometricLoss("facenet", n=1, seed=2), gamma=10.0, **kw)
    lit = mk(ProjectionEngine, batch=1, use_graph=False).run(1)
    first_literal = float(lit.losses[0])
    eng = mk(GradientProjectionEngine, use_graph=True).run()
    lat, bstep, bloss, losses = eng.result()
    assert np.isfinite(losses).all()
    assert abs(losses[0] - first_literal) < 1e-4 * abs(first_literal)       # step 0: lr = 0, same candidate, same objective
    assert float((eng.latent_in[0] - latent_mean).abs().max()) > 0.01
