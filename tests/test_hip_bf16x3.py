"""The OPT-IN "bf16x3" arithmetic (mgf_conv_taps_bf16x3_f32; never the default, never the headline): every float32 operand split into two
bfloat16 terms, three v_mfma_f32_32x32x16_bf16 per product, float32 accumulation.  Gates: against float64 torch the error stays below 1e-5 of
max|y| on the generator's layer shapes (the float32 kernel: ~1e-6); the float32 path is untouched."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF_TOL = 1e-5


def _rel(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("n,cin,cout,res,styled,ep", [(2, 64, 64, 64, True, True), (1, 32, 32, 128, True, False), (3, 128, 96, 40, False, True),
                                                       (1, 512, 512, 32, True, True), (2, 16, 32, 96, True, True)])
def test_conv3x3_bf16x3_vs_float64(n, cin, cout, res, styled, ep):
    """3x3 / stride 1 / pad 1 with style on the input channels, demodulation on the output channels and the fused noise / bias / lrelu /
    residual epilogue -- networks.py:253-328's modulated convolution -- in the bf16x3 arithmetic against float64, beside the float32 kernel."""
    from morphganformer_amd import _lib
    from morphganformer_amd import conv as cv
    torch.manual_seed(cin + cout + res)
    x = torch.randn(n, cin, res, res)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    s = torch.rand(n, cin) + 0.5 if styled else None
    dm = torch.rand(n, cout) + 0.5 if styled else None
    bias, noise, resid = torch.randn(cout), torch.randn(n, res * res), torch.randn(n, cout, res, res)
    pc = cv.pack_weights(w.cuda())
    wb = cv.pack_weights_bf16x3(pc)
    mk = lambda: _lib.make_epilogue(bias=bias.cuda(), noise=noise.cuda(), noise_strength=torch.tensor([0.3]).cuda(), noise_n=n, act="lrelu", alpha=0.2,
                                    gain=1.2, residual=resid.cuda()) if ep else None
    keep = [bias, noise, resid]
    d = lambda t: None if t is None else t.cuda()
    y32 = cv.conv_forward(x.cuda(), pc, pad=(1, 1), in_scale=d(s), out_scale=d(dm), epilogue=mk())
    ybf = cv.conv_forward(x.cuda(), pc, pad=(1, 1), in_scale=d(s), out_scale=d(dm), epilogue=mk(), bf=wb)
    xd = x.double() * (s.double()[:, :, None, None] if styled else 1.0)
    ref = torch.nn.functional.conv2d(xd, w.double(), padding=1)
    if styled:
        ref = ref * dm.double()[:, :, None, None]
    if ep:
        ref = ref + 0.3 * noise.double().view(n, 1, res, res) + bias.double().view(1, -1, 1, 1)
        ref = torch.where(ref > 0, ref, 0.2 * ref) * 1.2 + resid.double()
    assert _rel(y32, ref) < 3e-6
    assert _rel(ybf, ref) < BF_TOL, _rel(ybf, ref)
    assert not torch.equal(ybf, y32)                      # (it IS another arithmetic)


@pytest.mark.parametrize("n,cin,cout,res", [(4, 64, 32, 64), (1, 128, 64, 128), (4, 512, 512, 32), (5, 32, 32, 96)])
def test_transposed_conv_bf16x3_vs_float64(n, cin, cout, res):
    """The stride-2 transposed 3x3 conv of the up-sampling layers (conv2d_resample.py:118-131) -- 9 taps into 4 output parities -- in the
    bf16x3 arithmetic (the main launch; the last row / column stay on the float32 border kernel) against float64."""
    from morphganformer_amd import conv as cv
    torch.manual_seed(cin * res)
    x = torch.randn(n, cin, res, res)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    s, dm = torch.rand(n, cin) + 0.5, torch.rand(n, cout) + 0.5
    pc = cv.pack_weights(w.cuda())
    wb = cv.pack_weights_bf16x3(pc)
    t32 = cv.tconv3x3s2_forward(x.cuda(), pc, in_scale=s.cuda(), out_scale=dm.cuda()).clone()
    tbf = cv.tconv3x3s2_forward(x.cuda(), pc, in_scale=s.cuda(), out_scale=dm.cuda(), bf=wb).clone()
    ref = torch.nn.functional.conv_transpose2d(x.double() * s.double()[:, :, None, None], w.double().permute(1, 0, 2, 3), stride=2)
    ref = ref * dm.double()[:, :, None, None]
    assert tuple(tbf.shape) == tuple(ref.shape)
    assert _rel(t32, ref) < 3e-6
    assert _rel(tbf, ref) < BF_TOL, _rel(tbf, ref)
    if n >= 4 or res >= 128:
        assert not torch.equal(tbf, t32)                  # the split form ran (smaller launches keep the float32 single launch)


def test_bf16x3_refuses_what_it_does_not_serve():
    from morphganformer_amd import _lib
    from morphganformer_amd import conv as cv
    w = torch.randn(32, 16, 3, 3).cuda()
    pc = cv.pack_weights(w)
    wb = cv.pack_weights_bf16x3(pc)
    with pytest.raises(_lib.MgfError, match="bf16x3"):
        cv.conv_forward(torch.randn(1, 16, 8, 8).cuda(), pc, pad=(1, 1), bf=wb)            # an 8 x 8 map: the small-tile geometry
    with pytest.raises(_lib.MgfError, match="bf16x3"):
        cv.conv_forward(torch.randn(1, 16, 64, 64).cuda(), pc, stride=2, pad=(1, 1), bf=wb)  # stride 2


def test_generator_bf16x3_mode_at_full_size(golden):
    """Generator(arith="bf16x3") at 1024^2, 8 candidates per forward (the batched dispatch: transposed convs from 32^2 up and the 3x3 layers of
    the 64^2 / 128^2 blocks in the bf16x3 arithmetic): the reference module's own 1024^2 output (tests/golden/gen_full1024.npz) to the north
    star's 1e-3 on the sampled pixels -- measured far inside it -- and every pixel against the float32 engine to 1e-4; the default engine is
    untouched (arith="f32")."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    g = golden("gen_full1024.npz")
    cfg, B = FULL1024, 8
    sd = make_state_dict(cfg, seed=0)
    z = torch.cat([torch.from_numpy(g["z"]), torch.from_numpy(synthetic_latents(cfg, B - 1, 77))]).cuda()
    G32 = Generator(sd, cfg, "cuda", max_batch=B)
    assert G32.arith == "f32" and all(lp.pcb is None for lp in G32.plan.layers)
    ref = G32(z, None, noise_mode="const")[0]
    del G32
    Gbf = Generator(sd, cfg, "cuda", max_batch=B, arith="bf16x3")
    assert sum(lp.pcb is not None for lp in Gbf.plan.layers) >= 8
    img = Gbf(z, None, noise_mode="const")[0]
    amax = float(g["img_absmax"])
    pix = img[0].reshape(-1)[torch.from_numpy(g["idx"]).cuda()].cpu().numpy()
    assert np.abs(pix - g["pixels"]).max() / amax < 1e-3
    err = float((img - ref).abs().max()) / float(ref.abs().max())
    assert 0 < err < 1e-4, err
    with pytest.raises(ValueError, match="arith"):
        Generator(sd, cfg, "cuda", arith="fp8")


def test_literal_loop_in_bf16x3_mode_selects_the_same_candidate():
    """The literal loop on a bf16x3 generator at 1024^2 (32 candidates per forward, MSE objective, injected eps): the best latent is the
    float32 run's -- it is a perturbation of the start latent, not a function of the image -- and the loss history agrees to 1e-4."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, latent_stats
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    cfg = FULL1024
    sd = make_state_dict(cfg, seed=0)
    out = {}
    for arith in ("f32", "bf16x3"):
        G = Generator(sd, cfg, "cuda", max_batch=1, arith=arith)
        if arith == "f32":
            target = G(torch.from_numpy(synthetic_latents(cfg, 1, 1000)).cuda(), None, noise_mode="const")[0].clamp(-1, 1).clone()
            gen = torch.Generator(device="cuda"); gen.manual_seed(0)
            mean, std = latent_stats(G, 10000, "cuda", gen)
            eps = torch.randn(32, 1, cfg.k, cfg.z_dim, device="cuda", generator=gen)
        eng = ProjectionEngine(G, target, mean, std, ProjectionArgs(step=32), percept=None, use_mse=True, eps=eps, noise_mode="const", batch=32)
        out[arith] = eng.run().result()
        del eng, G
    (lat_a, step_a, _, hist_a), (lat_b, step_b, _, hist_b) = out["f32"], out["bf16x3"]
    assert step_a == step_b and torch.equal(lat_a, lat_b)
    assert np.abs(hist_a - hist_b).max() <= 1e-4 * np.abs(hist_a).max()
