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
