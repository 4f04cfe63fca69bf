"""Gradient mode: d(loss)/d(latent) through the HIP generator against torch autograd through the CPU
oracle (oracle/generator_ref.py), which tests/test_oracle_golden.py pins on the reference module's own autograd
(the `grad_z` vector of tests/golden/gen_tiny.npz).  Gate: max |difference| <= 1e-3 of max |gradient| (the north-star pixel tolerance, applied to
gradients)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-3


def rel(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def tiny():
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.grad import GeneratorGrad
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.generator_ref import to_torch_state
    sd = make_state_dict(TINY, seed=0)
    G = Generator(sd, TINY, "cuda", max_batch=2)
    return GeneratorGrad(G), to_torch_state(sd), TINY


def test_mapping_backward_matches_autograd(tiny):
    from morphganformer_amd import _lib
    from oracle.generator_ref import mapping_ref
    gg, tsd, cfg = tiny
    torch.manual_seed(5)
    z = torch.randn(3, cfg.k, cfg.z_dim, requires_grad=True)
    dw = torch.randn(3, cfg.k, cfg.w_dim)
    w = mapping_ref(tsd, z, cfg)
    (ref,) = torch.autograd.grad(w, z, dw)
    L = _lib.lib()
    zc, dwc = z.detach().cuda(), dw.cuda()
    dz = torch.empty_like(zc)
    scratch = torch.empty(3 * int(L.mgf_mapping_bwd_scratch_floats(cfg.k, cfg.w_dim, cfg.mapping_layers // 2)), device="cuda")
    _lib.check(L.mgf_mapping_backward(dz.data_ptr(), dwc.data_ptr(), zc.data_ptr(), gg.G.plan.mapping_blob.data_ptr(), scratch.data_ptr(),
                                      3, cfg.k, cfg.w_dim, cfg.mapping_layers // 2, int(cfg.normalize_global), _lib.stream_ptr()))
    assert rel(dz, ref) < GRAD_TOL
    # the saving forward + the backward that reads its slab: the same w as the plain forward and the same dz, bit for bit
    scratch2 = torch.full_like(scratch, float("nan"))
    w0, w1, dz2 = torch.empty_like(zc), torch.empty_like(zc), torch.empty_like(zc)
    margs = (3, cfg.k, cfg.w_dim, cfg.mapping_layers // 2, int(cfg.normalize_global), _lib.stream_ptr())
    _lib.check(L.mgf_mapping_forward(w0.data_ptr(), zc.data_ptr(), gg.G.plan.mapping_blob.data_ptr(), *margs))
    _lib.check(L.mgf_mapping_forward_save(w1.data_ptr(), zc.data_ptr(), gg.G.plan.mapping_blob.data_ptr(), scratch2.data_ptr(), *margs))
    _lib.check(L.mgf_mapping_backward_saved(dz2.data_ptr(), dwc.data_ptr(), zc.data_ptr(), gg.G.plan.mapping_blob.data_ptr(),
                                            scratch2.data_ptr(), *margs))
    assert torch.equal(w0, w1) and torch.equal(dz, dz2)


@pytest.mark.parametrize("k,layers", [(6, 4), (12, 14), (33, 2), (2, 0)])
def test_mapping_forward_and_backward_other_shapes(k, layers):
    """Component counts that leave the last pass of the [T x D] products ragged (T = 5, 11), the largest T (32), the smallest
    (1), and residual-layer counts that fill one / both register sets of the global path's groups (2 n_res + 1 = 5, 15, 3, 1)."""
    import dataclasses
    from morphganformer_amd import _lib
    from morphganformer_amd.engine import pack_mapping_params
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.generator_ref import mapping_ref, to_torch_state
    cfg = dataclasses.replace(TINY, k=k, mapping_layers=layers)
    sd = make_state_dict(cfg, seed=3)
    blob = torch.from_numpy(pack_mapping_params(sd, cfg)).cuda()
    torch.manual_seed(k)
    z = torch.randn(2, k, cfg.z_dim, requires_grad=True)
    dw = torch.randn(2, k, cfg.w_dim)
    w_ref = mapping_ref(to_torch_state(sd), z, cfg)
    (dz_ref,) = torch.autograd.grad(w_ref, z, dw)
    L = _lib.lib()
    zc, dwc = z.detach().cuda(), dw.cuda()
    w0, w1, dz0, dz1 = (torch.empty_like(zc) for _ in range(4))
    scr = [torch.full((2 * int(L.mgf_mapping_bwd_scratch_floats(k, cfg.w_dim, layers // 2)),), float("nan"), device="cuda") for _ in range(2)]
    margs = (2, k, cfg.w_dim, layers // 2, int(cfg.normalize_global), _lib.stream_ptr())
    _lib.check(L.mgf_mapping_forward(w0.data_ptr(), zc.data_ptr(), blob.data_ptr(), *margs))
    _lib.check(L.mgf_mapping_forward_save(w1.data_ptr(), zc.data_ptr(), blob.data_ptr(), scr[0].data_ptr(), *margs))
    _lib.check(L.mgf_mapping_backward_saved(dz0.data_ptr(), dwc.data_ptr(), zc.data_ptr(), blob.data_ptr(), scr[0].data_ptr(), *margs))
    _lib.check(L.mgf_mapping_backward(dz1.data_ptr(), dwc.data_ptr(), zc.data_ptr(), blob.data_ptr(), scr[1].data_ptr(), *margs))
    assert rel(w0, w_ref) < 1e-5 and torch.equal(w0, w1)
    assert rel(dz0, dz_ref) < GRAD_TOL and torch.equal(dz0, dz1)
    assert L.mgf_mapping_forward(w0.data_ptr(), zc.data_ptr(), blob.data_ptr(), 2, k, cfg.w_dim, 8, 1, _lib.stream_ptr()) != 0      # > 7 res layers


def test_layer_act_bwd_and_dots():
    from morphganformer_amd import _lib
    L = _lib.lib()
    torch.manual_seed(1)
    n, c, hw = 2, 5, 5000                               # two chunks per plane, ragged tail
    cpre = torch.randn(n, c, hw, dtype=torch.float64)
    noise = torch.randn(n, hw, dtype=torch.float64)
    bias = torch.randn(c, dtype=torch.float64)
    res = torch.randn(n, c, hw, dtype=torch.float64)
    dy = torch.randn(n, c, hw, dtype=torch.float64)
    ns, gain = 0.3, 0.9
    z = cpre + noise[:, None] * ns + bias[None, :, None]
    y = torch.nn.functional.leaky_relu(z, 0.2) * gain + res
    dz_ref = dy * gain * torch.where(z > 0, 1.0, 0.2)
    dot_ref = (dz_ref * cpre).sum(-1)
    f = lambda t: t.float().cuda().contiguous()
    yd, rd, dyd, nd, bd = f(y), f(res), f(dy), f(noise), f(bias)
    nsd = torch.tensor([ns], device="cuda")
    chunks = int(L.mgf_bwd_chunks(hw))
    assert chunks == 2
    dz = torch.empty_like(yd)
    part = torch.empty(n, c, chunks, device="cuda")
    _lib.check(L.mgf_layer_act_bwd_f32(dz.data_ptr(), part.data_ptr(), dyd.data_ptr(), yd.data_ptr(), rd.data_ptr(), bd.data_ptr(),
                                       nd.data_ptr(), nsd.data_ptr(), n, n, c, hw, 0.2, gain, _lib.stream_ptr()))
    # elements whose pre-activation is within float32 noise of zero may take either slope
    decided = (z.abs() > 1e-5).cuda()
    assert torch.allclose(dz[decided], f(dz_ref)[decided], rtol=1e-5, atol=1e-6)
    assert rel(part.sum(-1), dot_ref) < 1e-4
    # channel_dot / style_grad
    a, b, s = f(torch.randn(n, c, hw)), f(torch.randn(n, c, hw)), f(torch.randn(n, c))
    _lib.check(L.mgf_channel_dot_f32(part.data_ptr(), a.data_ptr(), b.data_ptr(), n, c, hw, _lib.stream_ptr()))
    assert rel(part.sum(-1), (a.double() * b.double()).sum(-1)) < 1e-5
    dx = torch.full_like(a, 2.0)
    _lib.check(L.mgf_style_grad_f32(part.data_ptr(), dx.data_ptr(), a.data_ptr(), b.data_ptr(), s.data_ptr(), n, c, hw, 1, _lib.stream_ptr()))
    assert rel(part.sum(-1), (a.double() * b.double()).sum(-1)) < 1e-5
    assert torch.allclose(dx, 2.0 + s[:, :, None] * b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("hw,with_dot,with_res", [(5000, True, False), (4096, False, False), (37, True, False),
                                                  (5000, True, True), (4096, False, True), (37, True, True)])
def test_style_grad_fused_with_act_bwd_is_the_two_kernels_in_sequence(hw, with_dot, with_res):
    """mgf_style_grad_act_bwd_f32 (conv1's style gradient + conv0's activation backward in one pass; with a residual and the stored s g:
    conv_last's style gradient + conv1's activation backward) == mgf_style_grad_f32 followed by mgf_layer_act_bwd_f32, bit for bit:
    same partial sums, same dz, same dx."""
    from morphganformer_amd import _lib
    L = _lib.lib()
    torch.manual_seed(hw)
    n, c = 2, 5
    y = torch.randn(n, c, hw, device="cuda")                       # conv0's output = conv1's input
    g = torch.randn(n, c, hw, device="cuda")
    s = torch.rand(n, c, device="cuda") + 0.5
    bias, noise = torch.randn(c, device="cuda"), torch.randn(n, hw, device="cuda")
    nstr = torch.tensor([0.3], device="cuda")
    res = torch.randn(n, c, hw, device="cuda") if with_res else None
    rp = res.data_ptr() if with_res else None
    chunks = int(L.mgf_bwd_chunks(hw))
    st = _lib.stream_ptr()
    ps_a, pd_a = torch.empty(n, c, chunks, device="cuda"), torch.empty(n, c, chunks, device="cuda")
    dmid, dz_a = torch.empty_like(y), torch.empty_like(y)
    _lib.check(L.mgf_style_grad_f32(ps_a.data_ptr(), dmid.data_ptr(), y.data_ptr(), g.data_ptr(), s.data_ptr(), n, c, hw, 0, st))
    _lib.check(L.mgf_layer_act_bwd_f32(dz_a.data_ptr(), pd_a.data_ptr() if with_dot else None, dmid.data_ptr(), y.data_ptr(), rp,
                                       bias.data_ptr(), noise.data_ptr(), nstr.data_ptr(), n, n, c, hw, 0.2, 1.3, st))
    ps_b, pd_b, dz_b, dx_b = torch.empty_like(ps_a), torch.empty_like(pd_a), torch.empty_like(y), torch.empty_like(y)
    _lib.check(L.mgf_style_grad_act_bwd_f32(ps_b.data_ptr(), pd_b.data_ptr() if with_dot else None, dz_b.data_ptr(),
                                            dx_b.data_ptr() if with_res else None, y.data_ptr(), g.data_ptr(), s.data_ptr(), rp, None, 0,
                                            bias.data_ptr(), noise.data_ptr(), nstr.data_ptr(), n, n, c, hw, 0.2, 1.3, st))
    assert torch.equal(dz_a, dz_b) and torch.equal(ps_a, ps_b)
    if with_res:
        assert torch.equal(dmid, dx_b)
        assert L.mgf_style_grad_act_bwd_f32(ps_b.data_ptr(), None, dz_b.data_ptr(), dx_b.data_ptr(), y.data_ptr(), g.data_ptr(), s.data_ptr(),
                                            None, None, 0, None, None, None, 0, n, c, hw, 0.2, 1.3, st) != 0  # dx without residual
    if with_dot:
        assert torch.equal(pd_a, pd_b)


@pytest.mark.parametrize("h,w", [(8, 8), (6, 12), (64, 128)])
def test_activation_backward_with_the_residual_at_half_resolution(h, w):
    """The skip branch consumed at HALF resolution (the forward's Winograd epilogue up-samples it: engine fuse_skip_up): both activation
    backward kernels up-sample it themselves -- same dz / dx / partial sums as with the full-resolution tensor that upfirdn2d(up=2, pad
    (2,1,2,1), gain 4) makes of it (to rounding: another summation order inside the 4-tap interpolation)."""
    from morphganformer_amd import _lib, conv as cv
    from morphganformer_amd.torch_utils.ops import upfirdn2d
    L = _lib.lib()
    torch.manual_seed(h * w)
    n, c, hw = 2, 5, h * w
    low = torch.randn(n, c, h // 2, w // 2, device="cuda")
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
    full = torch.empty(n, c, h, w, device="cuda")
    cv.upfirdn_into(full, low, f, up=2, pad=(2, 1, 2, 1), gain=4.0)
    ref_up = torch.nn.functional.conv_transpose2d(low.double().cpu().reshape(n * c, 1, h // 2, w // 2),
                                                  (torch.outer(torch.tensor([1., 3, 3, 1]), torch.tensor([1., 3, 3, 1])).double() / 16)[None, None],
                                                  stride=2, padding=1).reshape(n, c, h, w)
    assert rel(full, ref_up) < 1e-6                              # (what "up-sampled" means here)
    y = torch.randn(n, c, h, w, device="cuda") + full
    dy, g = torch.randn_like(y), torch.randn_like(y)
    s = torch.rand(n, c, device="cuda") + 0.5
    bias, noise, nstr = torch.randn(c, device="cuda"), torch.randn(n, hw, device="cuda"), torch.tensor([0.3], device="cuda")
    chunks = int(L.mgf_bwd_chunks(hw))
    st = _lib.stream_ptr()
    e = lambda: (torch.empty_like(y), torch.empty(n, c, chunks, device="cuda"))
    (dz_a, pd_a), (dz_b, pd_b) = e(), e()
    _lib.check(L.mgf_layer_act_bwd_low_f32(dz_a.data_ptr(), pd_a.data_ptr(), dy.data_ptr(), y.data_ptr(), full.data_ptr(), None, 0, bias.data_ptr(),
                                           noise.data_ptr(), nstr.data_ptr(), n, n, c, hw, 0.2, 1.3, st))
    _lib.check(L.mgf_layer_act_bwd_low_f32(dz_b.data_ptr(), pd_b.data_ptr(), dy.data_ptr(), y.data_ptr(), None, low.data_ptr(), w, bias.data_ptr(),
                                           noise.data_ptr(), nstr.data_ptr(), n, n, c, hw, 0.2, 1.3, st))
    decided = (y - full).abs() > 1e-5                              # (an element at the kink may take either slope)
    assert torch.equal(dz_a[decided], dz_b[decided]) and rel(pd_a.sum(-1), pd_b.sum(-1)) < 1e-5
    (dz_c, pd_c), (dz_d, pd_d) = e(), e()
    ps_c, ps_d, dx_c, dx_d = torch.empty_like(pd_c), torch.empty_like(pd_c), torch.empty_like(y), torch.empty_like(y)
    for dzv, pdv, psv, dxv, full_p, low_p, ww in ((dz_c, pd_c, ps_c, dx_c, full.data_ptr(), None, 0), (dz_d, pd_d, ps_d, dx_d, None, low.data_ptr(), w)):
        _lib.check(L.mgf_style_grad_act_bwd_f32(psv.data_ptr(), pdv.data_ptr(), dzv.data_ptr(), dxv.data_ptr(), y.data_ptr(), g.data_ptr(),
                                                s.data_ptr(), full_p, low_p, ww, bias.data_ptr(), noise.data_ptr(), nstr.data_ptr(), n, n, c, hw,
                                                0.2, 1.3, st))
    assert torch.equal(dz_c[decided], dz_d[decided]) and torch.equal(dx_c, dx_d) and torch.equal(ps_c, ps_d)
    assert rel(pd_c.sum(-1), pd_d.sum(-1)) < 1e-5
    # both at once / an odd width are refused
    assert L.mgf_layer_act_bwd_low_f32(dz_a.data_ptr(), None, dy.data_ptr(), y.data_ptr(), full.data_ptr(), low.data_ptr(), w, None, None, None, 0,
                                       n, c, hw, 0.2, 1.3, st) != 0
    assert L.mgf_layer_act_bwd_low_f32(dz_a.data_ptr(), None, dy.data_ptr(), y.data_ptr(), None, low.data_ptr(), w - 1, None, None, None, 0,
                                       n, c, hw, 0.2, 1.3, st) != 0


@pytest.mark.parametrize("c,res", [(32, 8), (512, 4), (20, 6), (256, 32), (64, 16), (512, 8), (256, 8)])
def test_duplex_attention_bwd_matches_autograd(c, res):
    from morphganformer_amd import _lib
    L = _lib.lib()
    torch.manual_seed(c)
    n, T, f = 2, 16, res * res
    x = torch.randn(n, c, f, dtype=torch.float64, requires_grad=True)
    wqc = torch.randn(c, T, dtype=torch.float64) / math.sqrt(c)
    spos = torch.randn(f, T, dtype=torch.float64)
    vwb = (1 + 0.3 * torch.randn(n, c, T, dtype=torch.float64)).requires_grad_(True)
    da = torch.randn(n, c, f, dtype=torch.float64)
    S = torch.einsum("ncf,ct->nft", x, wqc) + spos[None]
    P = torch.softmax(S, -1)
    r = torch.rsqrt(x.square().mean(1, keepdim=True) + 1e-8)
    g = torch.einsum("nft,nct->ncf", P, vwb)
    a = x * r * g
    dx_ref, dv_ref = torch.autograd.grad(a, (x, vwb), da)
    fl = lambda t: t.detach().float().cuda().contiguous()
    xd, wq, sp, vw, dad = fl(x), fl(wqc), fl(spos), fl(vwb), fl(da)
    dx, dg, probs = torch.empty_like(xd), torch.empty_like(xd), torch.empty(n, f, T, device="cuda")
    _lib.check(L.mgf_duplex_attention_bwd(dx.data_ptr(), dg.data_ptr(), probs.data_ptr(), dad.data_ptr(), xd.data_ptr(), wq.data_ptr(),
                                          sp.data_ptr(), vw.data_ptr(), n, c, f, T, _lib.stream_ptr()))
    dv = torch.empty(n, c, T, device="cuda")
    _lib.check(L.mgf_attn_values_grad(dv.data_ptr(), dg.data_ptr(), probs.data_ptr(), n, c, f, T, _lib.stream_ptr()))
    assert rel(probs, P) < 1e-4
    assert rel(dx, dx_ref) < GRAD_TOL
    assert rel(dv, dv_ref) < GRAD_TOL
    # the MFMA form (pixel slices + fixed-order reduce) of the value gradient: same numbers, bit-reproducible
    ws = torch.empty(int(L.mgf_attn_values_grad_workspace_floats(n, c)), device="cuda")
    dv2, dv3 = torch.full_like(dv, 7.0), torch.full_like(dv, 9.0)
    for out in (dv2, dv3):
        _lib.check(L.mgf_attn_values_grad_ws(out.data_ptr(), dg.data_ptr(), probs.data_ptr(), n, c, f, T, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    assert rel(dv2, dv_ref) < GRAD_TOL and torch.equal(dv2, dv3)


@pytest.mark.parametrize("mode", ["const", "inject"])
def test_tiny_synthesis_gradient_wrt_w(tiny, mode):
    from oracle.generator_ref import mapping_ref, synthesis_ref
    gg, tsd, cfg = tiny
    torch.manual_seed(11)
    z = torch.randn(2, cfg.k, cfg.z_dim)
    dimg = torch.randn(2, 3, cfg.img_resolution, cfg.img_resolution)
    noises_cpu = noises = None
    if mode == "inject":
        noises_cpu = {lp.name: torch.randn(2, lp.res, lp.res) for lp in gg.G.plan.layers if lp.noise_strength is not None}
        noises = {k: v.cuda().reshape(2, -1) for k, v in noises_cpu.items()}
    w = mapping_ref(tsd, z, cfg).detach().requires_grad_(True)
    img_ref = synthesis_ref(tsd, w, cfg, mode, noises_cpu)
    (dw_ref,) = torch.autograd.grad(img_ref, w, dimg)
    ws = w.detach().cuda().unsqueeze(2).expand(-1, -1, cfg.num_ws, -1)
    img = gg.forward(ws=ws, noise_mode=mode, noises=noises)
    assert rel(img, img_ref) < 1e-3
    dw = gg.backward_w(dimg.cuda())
    assert rel(dw, dw_ref) < GRAD_TOL


def test_tiny_generator_gradient_wrt_z(tiny):
    from oracle.generator_ref import generator_ref
    gg, tsd, cfg = tiny
    torch.manual_seed(12)
    z = torch.randn(2, cfg.k, cfg.z_dim, requires_grad=True)
    target = torch.randn(2, 3, cfg.img_resolution, cfg.img_resolution) * 0.5
    img_ref = generator_ref(tsd, z, cfg, "const")
    loss = (img_ref - target).square().mean(dim=(1, 2, 3)).sum()
    (dz_ref,) = torch.autograd.grad(loss, z)
    img = gg.forward(z.detach().cuda(), noise_mode="const")
    dimg = 2.0 * (img - target.cuda()) / target[0].numel()
    dz = gg.backward(dimg)
    assert rel(dz, dz_ref) < GRAD_TOL
    # batch of one re-allocates the workspaces and must agree with the batched run
    img1 = gg.forward(z.detach()[:1].cuda(), noise_mode="const")
    dz1 = gg.backward(2.0 * (img1 - target[:1].cuda()) / target[0].numel())
    assert rel(dz1, dz_ref[:1]) < GRAD_TOL


def test_deferred_attention_by_products_equal_the_per_layer_launches(monkeypatch):
    """The attention layers' value gradients and demodulation dot products as two launches at the end of the pass (mgf_attn_grad_multi: job
    table over all layers, per-layer buffers) against the three small launches per layer where they arise: the same arithmetic in the same
    order -- dz bit-identical -- at one and at three samples, and again after a workspace of another batch size was used in between."""
    from morphganformer_amd import grad
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    sd = make_state_dict(TINY, seed=0)
    torch.manual_seed(5)
    out = {}
    for defer in (True, False):
        monkeypatch.setattr(grad, "DEFER_ATTN_GRADS", defer)
        gg = grad.GeneratorGrad(Generator(sd, TINY, "cuda", max_batch=1))
        res = []
        for n in (1, 3, 1):
            torch.manual_seed(n)
            z = torch.randn(n, TINY.k, TINY.z_dim, device="cuda")
            img = gg.forward(z, noise_mode="const")
            res.append(gg.backward(torch.sin(img * 3.0)).clone())
            assert (gg.attn_defer is not None) == defer
        out[defer] = res
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)
    assert torch.equal(out[True][0], out[True][2])


@pytest.mark.parametrize("res,cmax,attn", [(256, 512, 8), (128, 64, 5)])
def test_fused_style_act_blur_gradient_pass_equals_the_separate_launches(monkeypatch, res, cmax, attn):
    """Up-sampling layers without attention: conv1's style gradient, conv0's activation backward and the adjoint of conv0's blur as ONE pass
    (mgf_style_act_fir_bwd_f32: conv0's dz never reaches memory, dot-product partials per 64 x 64 tile) against the two launches it replaces
    (mgf_style_grad_act_bwd_f32 + upfirdn2d with pad 2): d/dz to 1e-5 -- the same arithmetic per element, another summation order of the
    partials -- at one and at two samples, random per-layer noise, on the 256^2 generator (its top block) and on a 128^2 one whose blocks from
    32^2 up have no attention (three fused blocks, 32^2 / 64^2 / 128^2 maps: tiles cut by the map edge)."""
    from morphganformer_amd import grad
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import GeneratorConfig, make_state_dict
    cfg = GeneratorConfig(img_resolution=res, channel_max=cmax, attn_max_log2res=attn)
    sd = make_state_dict(cfg, seed=0)
    out = {}
    for fuse in (True, False):
        monkeypatch.setattr(grad, "FUSE_ACT_FIR", fuse)
        G = Generator(sd, cfg, "cuda", max_batch=1)
        gg = grad.GeneratorGrad(G)
        res_ = []
        for n in (1, 2):
            torch.manual_seed(n)
            z = torch.randn(n, cfg.k, cfg.z_dim, device="cuda")
            noises = {lp.name: torch.randn(n, lp.res * lp.res, device="cuda") for lp in G.plan.layers if lp.noise_strength is not None}
            img = gg.forward(z, noise_mode="inject", noises=noises)
            res_.append(gg.backward(torch.sin(img * 3.0)).clone())
            assert gg._fir_mode == fuse
        out[fuse] = res_
    for a, b in zip(out[True], out[False]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())


def test_tiny_gradient_matches_reference_module(tiny, golden):
    """d mean(img^2)/dz computed by the REFERENCE module's autograd (gen_tiny.npz, oracle/make_golden.py) vs the HIP backward."""
    gg, tsd, cfg = tiny
    g = golden("gen_tiny.npz")
    z = torch.from_numpy(g["z"]).cuda()
    img = gg.forward(z, noise_mode="const")
    assert abs(float(img.square().mean()) - float(g["loss_sq"])) < 1e-3 * float(g["loss_sq"])
    dz = gg.backward(2.0 * img / img.numel())
    assert rel(dz, g["grad_z"]) < GRAD_TOL


def test_wplus_gradient_matches_reference_module(tiny, golden):
    """W+ (per-layer latents, networks.py:1252-1253,1304-1331): d mean(img^2) / d ws [n, k, num_ws, D] by the REFERENCE module's
    autograd (tests/golden/wplus_tiny.npz: grad_ws) vs GeneratorGrad.backward_ws -- every layer's style / attention-value gradient in
    its own slot.  Summed over the slots it is backward_w(); a dense broadcast ws (what G.mapping(z, psi) / return_ws=True hand out)
    is accepted and gives the shared-latent gradient."""
    gg, tsd, cfg = tiny
    g = golden("wplus_tiny.npz")
    ws = torch.from_numpy(g["ws"]).cuda()
    img = gg.forward(ws=ws, noise_mode="const")
    assert rel(img, g["img_ws"]) < 1e-3
    assert abs(float(img.square().mean()) - float(g["loss_ws"])) < 1e-3 * float(g["loss_ws"])
    dimg = 2.0 * img / img.numel()
    dws = gg.backward_ws(dimg).clone()
    assert tuple(dws.shape) == tuple(g["grad_ws"].shape)
    assert rel(dws, g["grad_ws"]) < GRAD_TOL
    # per slot, not only overall: every layer slot's own gradient within tolerance of that slot's scale
    for s in range(cfg.num_ws):
        assert rel(dws[:, :, s], g["grad_ws"][:, :, s]) < 5 * GRAD_TOL, s
    dw = gg.backward_w(dimg)
    assert rel(dw, g["grad_ws"].sum(axis=2)) < GRAD_TOL
    # a MATERIALISED broadcast (dense copy of mapping()'s view) is the ordinary shared-latent case
    gt = golden("gen_tiny.npz")
    z = torch.from_numpy(gt["z"]).cuda()
    wsb = gg.G.mapping(z).contiguous()
    assert wsb.stride(2) != 0
    img_b = gg.forward(ws=wsb, noise_mode="const")
    dw_b = gg.backward_w(2.0 * img_b / img_b.numel()).clone()
    img_v = gg.forward(ws=gg.G.mapping(z), noise_mode="const")                     # the zero-stride view: shared tables
    dw_v = gg.backward_w(2.0 * img_v / img_v.numel())
    assert rel(img_b, img_v) < 1e-6 and rel(dw_b, dw_v) < 1e-4


@pytest.mark.parametrize("net,size", [("squeeze", 65), ("squeeze", 128), ("vgg", 48), ("vgg", 70), ("alex", 96), ("alex", 131), ("alex", 224)])
def test_lpips_gradient_matches_autograd(net, size):
    from morphganformer_amd.lpips import PerceptualLoss, random_backbone, WEIGHTS_DIR
    from oracle.loss_ref import backbone_random, lpips_ref
    import os
    torch.manual_seed(size)
    n = 2
    pred = (torch.rand(n, 3, size, size) * 2 - 1).requires_grad_(True)
    target = torch.rand(1, 3, size, size) * 2 - 1
    bb = backbone_random(net, 0)
    lin = np.load(os.path.join(WEIGHTS_DIR, f"lpips_lin_{net}.npz"))
    lins = [torch.from_numpy(lin[f"lin{i}"]).float().reshape(-1) for i in range(len(lin.files))]
    val = lpips_ref(bb, lins, pred, target.expand(n, -1, -1, -1), net=net)
    (ref,) = torch.autograd.grad(val.sum() * 0.7, pred)
    pl = PerceptualLoss(net=net, allow_random_backbone=True)
    pl.set_target(target.cuda())
    out = torch.empty(n, device="cuda")
    pl.distance_into(out, pred.detach().cuda(), keep_taps=True)
    assert rel(out, val.reshape(n)) < 1e-4
    dimg = torch.full((n, 3, size, size), 0.25, device="cuda")
    pl.grad_into(dimg, scale=0.7, accumulate=True)
    assert rel(dimg - 0.25, ref) < GRAD_TOL
    pl.grad_into(dimg, scale=0.7)
    assert rel(dimg, ref) < GRAD_TOL


def test_lpips_squeeze_gradient_with_winograd_data_gradients(monkeypatch):
    """The expand-3x3 data gradients of the Fire modules whose squeeze width is a multiple of 32 run on the Winograd kernel once the
    launch fills the chip (8 targets at 1024^2; odd 127^2 and 63^2 maps, residual added in place).  Forced here at a small size: the same
    backward from the same forward with and without them (against autograd a near-tie in a deep max-pool can differ between CPU
    and GPU forwards at an arbitrary size, which says nothing about these kernels; test_lpips_gradient_matches_autograd pins the rest)."""
    from morphganformer_amd import conv as cv
    from morphganformer_amd.lpips import PerceptualLoss
    monkeypatch.setattr(cv, "winograd_fills_chip", lambda *a: True)
    torch.manual_seed(11)
    n, size = 2, 140                                               # maps 69, 34, 17 (odd) and 8
    pred = torch.rand(n, 3, size, size, device="cuda") * 2 - 1
    pl = PerceptualLoss(net="squeeze", allow_random_backbone=True)
    pl.set_target(torch.rand(1, 3, size, size, device="cuda") * 2 - 1)
    out = torch.empty(n, device="cuda")
    pl.distance_into(out, pred, keep_taps=True)
    g_wino, g_taps = torch.empty_like(pred), torch.empty_like(pred)
    pl.grad_into(g_wino, scale=1.0)
    feat = pl._features(n, size, size)
    assert sorted(feat.gpw) == [6, 7, 11, 12]                      # Fire modules with 32 / 64 squeeze channels
    feat.gpw.clear()
    pl.grad_into(g_taps, scale=1.0)
    assert float(g_taps.abs().max()) > 0 and rel(g_wino, g_taps) < 1e-5


@pytest.mark.parametrize("c,split,hw,behind", [(128, 64, 20000, True), (64, 64, 17000, True), (48, 16, 300, True), (512, 256, 49, False),
                                                 (100, 40, 5000, True), (130, 130, 4999, False)])
def test_lpips_tap_gradient_fused_with_relu_bwd_is_the_two_kernels_in_sequence(c, split, hw, behind):
    """mgf_lpips_layer_bwd_relu_f32 == mgf_lpips_layer_bwd_f32 (accumulating into the gradient from behind the tap) followed by
    mgf_relu_bwd_split_f32, bit for bit; the three pixel-block sizes (64 / 32 / 16 per workgroup), channels cached in registers or not,
    with and without a split, one target per sample."""
    from morphganformer_amd import _lib
    L = _lib.lib()
    torch.manual_seed(c + hw)
    n = 2
    f0 = torch.relu(torch.randn(n, c, hw, device="cuda"))          # a ReLU output: about half zeros
    f1 = torch.nn.functional.normalize(torch.rand(n, c, hw, device="cuda"), dim=1)
    lin = torch.rand(c, device="cuda")
    dy = torch.randn(n, c, hw, device="cuda")
    st = _lib.stream_ptr()
    acc = dy.clone()
    _lib.check(L.mgf_lpips_layer_bwd_f32(acc.data_ptr(), f0.data_ptr(), f1.data_ptr(), lin.data_ptr(), n, c, hw, c * hw, 0.7, int(behind), st))
    a0, b0 = torch.empty(n, split, hw, device="cuda"), torch.empty(n, max(c - split, 1), hw, device="cuda")
    a1, b1 = torch.empty_like(a0), torch.empty_like(b0)
    bp = lambda t: t.data_ptr() if split < c else None
    _lib.check(L.mgf_relu_bwd_split_f32(a0.data_ptr(), bp(b0), acc.data_ptr(), f0.data_ptr(), n, c, split, hw, st))
    _lib.check(L.mgf_lpips_layer_bwd_relu_f32(a1.data_ptr(), bp(b1), dy.data_ptr() if behind else None, f0.data_ptr(), f1.data_ptr(),
                                              lin.data_ptr(), n, c, split, hw, c * hw, 0.7, st))
    assert torch.equal(a0, a1) and (split == c or torch.equal(b0, b1))
    # the per-pixel sums of the forward (mgf_lpips_layer_stats_f32) in place of the backward's own first sweep: the same sums in another
    # order, so equal to rounding; and the sums themselves against torch
    stats = torch.full((n, 3, hw), float("nan"), device="cuda")
    dist = torch.zeros(n, device="cuda")
    scratch = torch.empty(n * int(L.mgf_reduce_scratch_floats()), device="cuda")
    _lib.check(L.mgf_lpips_layer_stats_f32(dist.data_ptr(), stats.data_ptr(), f0.data_ptr(), f1.data_ptr(), lin.data_ptr(), n, c, hw, c * hw, 0,
                                           scratch.data_ptr(), st))
    want = torch.stack([(f0 * f0).sum(1), (lin[None, :, None] * f0 * f0).sum(1), (lin[None, :, None] * f1 * f0).sum(1)], dim=1)
    assert rel(stats, want) < 1e-5
    a2, b2 = torch.empty_like(a0), torch.empty_like(b0)
    _lib.check(L.mgf_lpips_layer_bwd_relu_stats_f32(a2.data_ptr(), bp(b2), dy.data_ptr() if behind else None, f0.data_ptr(), f1.data_ptr(),
                                                    lin.data_ptr(), stats.data_ptr(), n, c, split, hw, c * hw, 0.7, st))
    assert rel(a2, a0) < 1e-5 and (split == c or rel(b2, b0) < 1e-5)
    if split == c:                                                 # in place, as the sequential backbones call it
        _lib.check(L.mgf_lpips_layer_bwd_relu_f32(dy.data_ptr(), None, dy.data_ptr(), f0.data_ptr(), f1.data_ptr(), lin.data_ptr(), n, c, c, hw,
                                                  c * hw, 0.7, st))
        assert torch.equal(dy, a0) or not behind


def test_mse_grad_and_pool_bwd():
    from morphganformer_amd import _lib
    L = _lib.lib()
    torch.manual_seed(3)
    a, b = torch.randn(2, 3, 9, 9), torch.randn(1, 3, 9, 9)
    ad, bd = a.cuda(), b.cuda()
    d = torch.ones_like(ad)
    _lib.check(L.mgf_mse_grad_f32(d.data_ptr(), ad.data_ptr(), bd.data_ptr(), 2, 243, 0, 0.5, 1, _lib.stream_ptr()))
    assert torch.allclose(d.cpu(), 1 + 0.5 * 2 * (a - b) / 243, rtol=1e-6, atol=1e-7)
    # ceil-mode pooling with ties (ReLU zeros): the gradient must follow torch's first-maximum rule
    # (one tile; several 64 x 32 tiles with ragged edges, odd and even sides; all-negative inputs: the -inf padding must never win)
    for shape, neg in (((2, 4, 15, 12), False), ((1, 3, 70, 131), False), ((2, 1, 65, 64), False), ((1, 2, 33, 129), True)):
        x = torch.relu(torch.randn(*shape))
        x = (x - 5.0 if neg else x).requires_grad_(True)
        y = torch.nn.functional.max_pool2d(x, 3, 2, ceil_mode=True)
        dy = torch.randn_like(y)
        (ref,) = torch.autograd.grad(y, x, dy)
        xd, dyd = x.detach().cuda(), dy.cuda()
        dx = torch.full_like(xd, float("nan"))
        _lib.check(L.mgf_maxpool3x3s2_ceil_bwd_f32(dx.data_ptr(), dyd.data_ptr(), xd.data_ptr(), shape[0] * shape[1], shape[2], shape[3],
                                                   y.shape[2], y.shape[3], _lib.stream_ptr()))
        assert torch.equal(dx.cpu(), ref), shape
        # the forward that stores each window's winning tap + the backward that reads those: same y as the plain forward, same dx
        yd, yi = torch.empty_like(dyd), torch.empty_like(dyd)
        taps = torch.full(dyd.shape, 255, dtype=torch.uint8, device="cuda")
        pargs = (shape[0] * shape[1], shape[2], shape[3], y.shape[2], y.shape[3], _lib.stream_ptr())
        _lib.check(L.mgf_maxpool3x3s2_ceil_f32(yd.data_ptr(), xd.data_ptr(), *pargs))
        _lib.check(L.mgf_maxpool3x3s2_ceil_idx_f32(yi.data_ptr(), taps.data_ptr(), xd.data_ptr(), *pargs))
        assert torch.equal(yd, yi) and torch.equal(yi.cpu(), y.detach()) and int(taps.max()) <= 8
        dx2 = torch.full_like(xd, float("nan"))
        _lib.check(L.mgf_maxpool3x3s2_ceil_bwd_idx_f32(dx2.data_ptr(), dyd.data_ptr(), taps.data_ptr(), *pargs))
        assert torch.equal(dx2.cpu(), ref), shape
    # forward pooling, both entries, odd and even sides, wider than one wave's 128 columns: bit-exact
    for (c, hh, ww) in ((3, 15, 12), (2, 31, 301), (1, 8, 257), (5, 2, 3)):
        xf = torch.randn(2, c, hh, ww)
        want = torch.nn.functional.max_pool2d(xf, 3, 2, ceil_mode=True)
        got = torch.empty_like(want).cuda()
        _lib.check(L.mgf_maxpool3x3s2_ceil_f32(got.data_ptr(), xf.cuda().data_ptr(), 2 * c, hh, ww, want.shape[2], want.shape[3], _lib.stream_ptr()))
        assert torch.equal(got.cpu(), want)
        for k in (2, 3):
            if hh < k or ww < k:
                continue
            want = torch.nn.functional.max_pool2d(xf, k, 2)
            got = torch.empty_like(want).cuda()
            _lib.check(L.mgf_maxpool_s2_floor_f32(got.data_ptr(), xf.cuda().data_ptr(), 2 * c, hh, ww, k, _lib.stream_ptr()))
            assert torch.equal(got.cpu(), want)


def test_wplus_gradient_projection_matches_autograd_adam(tiny):
    """GradientProjectionEngine(latent_space="w+"): the parameter is the per-layer latent ws [k, num_ws, D] (north_star: "backprops into the
    k-component latent W+") -- noise, Adam and the best-of bookkeeping act on all k * num_ws * D numbers -- against torch autograd +
    torch.optim.Adam on ws through the CPU restatement's synthesis network: LPIPS + MSE, injected noise, hipGraph replay; the start is a
    [k, D] mean broadcast over the slots, which must come apart (every slot gets its own gradient)."""
    import os
    from morphganformer_amd.lpips import PerceptualLoss, WEIGHTS_DIR
    from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs
    from morphganformer_amd.synth_weights import synthetic_latents
    from oracle.generator_ref import generator_ref, mapping_ref, synthesis_ref
    from oracle.loss_ref import backbone_random, lpips_ref, mse_ref, projection_gradient_ref
    gg, tsd, cfg = tiny
    steps = 8
    rng = np.random.Generator(np.random.PCG64(14))
    w_mean = mapping_ref(tsd, torch.from_numpy(synthetic_latents(cfg, 1, 77)), cfg)[0].detach()              # [k, D], a point of w space
    eps = torch.from_numpy(rng.standard_normal((steps, 1, cfg.k, cfg.num_ws, cfg.w_dim)).astype(np.float32))
    target = generator_ref(tsd, torch.from_numpy(synthetic_latents(cfg, 1, 1001)), cfg, "const").clamp(-1, 1)
    args = ProjectionArgs(step=steps, lr=0.05, lr_rampup=0.2)
    w_std = float(w_mean.std()) * 4
    bb = backbone_random("squeeze", 0)
    lin = np.load(os.path.join(WEIGHTS_DIR, "lpips_lin_squeeze.npz"))
    lins = [torch.from_numpy(lin[f"lin{i}"]).float().reshape(-1) for i in range(7)]
    loss_fn = lambda i, img: lpips_ref(bb, lins, img, target).sum() + args.beta * mse_ref(img, target)
    start = w_mean[:, None, :].expand(cfg.k, cfg.num_ws, cfg.w_dim).contiguous()
    ref = projection_gradient_ref(lambda ws: synthesis_ref(tsd, ws, cfg, "const"), loss_fn, start, w_std, eps, steps, lr=args.lr,
                                  rampdown=args.lr_rampdown, rampup=args.lr_rampup)
    eng = GradientProjectionEngine(gg.G, target.cuda(), w_mean.cuda(), w_std, args, percept=PerceptualLoss(net="squeeze", allow_random_backbone=True),
                                   eps=eps.cuda(), noise_mode="const", use_graph=True, latent_space="w+")
    traj = []
    for i in range(steps):
        eng.run(1)
        traj.append(eng.latent_in.cpu().clone())
    lat, bstep, bloss, losses = eng.result()
    assert tuple(lat.shape) == (1, cfg.k, cfg.num_ws, cfg.w_dim) and bstep == ref[1]
    for i in range(steps):
        assert float((traj[i] - ref[4][i]).abs().max()) < 0.05 * args.lr * (i + 1), i
    assert np.abs(losses - np.array(ref[3])).max() < 1e-3 * np.abs(np.array(ref[3])).max()
    assert rel(lat, ref[0]) < 0.02
    spread = traj[-1][0].std(dim=1).max()                                # the slots started equal and moved apart
    assert float(spread) > 0.2 * args.lr


@pytest.mark.parametrize("use_graph", [False, True])
def test_gradient_projection_matches_autograd_adam(tiny, use_graph):
    """Gradient-mode loop vs torch autograd + torch.optim.Adam through the CPU oracle: LPIPS + lamda Wing + beta MSE, a skipped
    ("no face") step, injected noise streams."""
    import os
    from morphganformer_amd.lpips import PerceptualLoss, WEIGHTS_DIR
    from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs, synthetic_landmarks
    from morphganformer_amd.synth_weights import synthetic_latents
    from oracle.generator_ref import generator_ref
    from oracle.loss_ref import backbone_random, lpips_ref, mse_ref, projection_gradient_ref, wing_loss_ref
    gg, tsd, cfg = tiny
    steps = 10
    rng = np.random.Generator(np.random.PCG64(4))
    latent_mean = torch.from_numpy(rng.standard_normal((cfg.k, cfg.z_dim)).astype(np.float32))
    eps = torch.from_numpy(rng.standard_normal((steps, 1, cfg.k, cfg.z_dim)).astype(np.float32))
    target = generator_ref(tsd, torch.from_numpy(synthetic_latents(cfg, 1, 1001)), cfg, "const").clamp(-1, 1)
    lm_t, lm_s = synthetic_landmarks(steps, 64, 9)
    valid = np.ones(steps, np.int32)
    valid[3] = 0
    args = ProjectionArgs(step=steps, lr=0.05, lr_rampup=0.2)
    bb = backbone_random("squeeze", 0)
    lin = np.load(os.path.join(WEIGHTS_DIR, "lpips_lin_squeeze.npz"))
    lins = [torch.from_numpy(lin[f"lin{i}"]).float().reshape(-1) for i in range(7)]

    def loss_fn(i, img):
        if not valid[i]:
            return None
        w = wing_loss_ref(torch.from_numpy(lm_s[i]), torch.from_numpy(lm_t))
        return lpips_ref(bb, lins, img, target).sum() + args.lamda * w + args.beta * mse_ref(img, target)

    ref = projection_gradient_ref(lambda z: generator_ref(tsd, z, cfg, "const"), loss_fn, latent_mean, 1.0, eps, steps, lr=args.lr,
                                  rampdown=args.lr_rampdown, rampup=args.lr_rampup)
    pl = PerceptualLoss(net="squeeze", allow_random_backbone=True)
    eng = GradientProjectionEngine(gg.G, target.cuda(), latent_mean.cuda(), 1.0, args, percept=pl, lm_target=lm_t, lm_steps=lm_s,
                                   lm_valid=valid, eps=eps.cuda(), noise_mode="const", use_graph=use_graph)
    traj = []
    for i in range(steps):
        eng.run(1)
        traj.append(eng.latent_in.cpu().clone())
    lat, bstep, bloss, losses = eng.result()
    moved = float((ref[4][-1] - latent_mean).abs().max())
    assert moved > 5 * args.lr * 0.2, "the oracle run must actually move the latent"
    for i in range(steps):
        # Adam normalises the gradient, so an element whose gradient is ~0 can differ by a fraction of one lr step
        assert float((traj[i] - ref[4][i]).abs().max()) < 0.05 * args.lr * (i + 1), i
    got = np.array([v for v in losses if not np.isnan(v)])
    want = np.array([v for v in ref[3] if v is not None])
    assert np.isnan(losses[3]) and ref[3][3] is None
    assert np.abs(got - want).max() < 1e-3 * np.abs(want).max()
    assert bstep == ref[1]
    assert float((lat - ref[0]).abs().max()) < 0.05 * args.lr * steps


def test_full1024_gradient_matches_reference_module(golden):
    """Full size: d MSE(G(z), target)/dz at 1024^2 by the REFERENCE module's autograd (tests/golden/grad_full1024.npz,
    oracle/make_golden.py: gold_grad_full) vs the HIP forward + backward."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.grad import GeneratorGrad
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict
    from oracle.make_golden import grad_full_target
    g = golden("grad_full1024.npz")
    gg = GeneratorGrad(Generator(make_state_dict(FULL1024, seed=0), FULL1024, "cuda", max_batch=1))
    target = torch.from_numpy(grad_full_target(1024)).cuda()
    img = gg.forward(torch.from_numpy(g["z"]).cuda(), noise_mode="const")
    loss = float((img - target).square().mean())
    assert abs(loss - float(g["loss"])) < 1e-4 * float(g["loss"])
    dz = gg.backward(2.0 * (img - target) / img.numel())
    assert rel(dz, g["grad_z"]) < GRAD_TOL


@pytest.mark.parametrize("size", [112, 160])
def test_biometric_gradient_matches_autograd(size):
    """d(gamma * MSE(embed(pred), embed(target)))/d(pred) through the IResNet-18 embedder (+ bilinear resize) vs FLOAT64 autograd
    through oracle/embed_ref.py (pinned on the reference module by tests/golden/iresnet18.npz).

    The 18 PReLU layers make the gradient piecewise: a pre-activation within float32 rounding of zero takes the other slope, in
    the HIP path and in float32 torch alike (tools/bio_grad_debug.py: without such a flip both sit at ~2e-6 of max|g| from
    float64, with one they jump to 1e-3..1e-2 on the pixels under that unit's receptive field).  So the gate is on the bulk of the
    error distribution -- median and rms -- not on its maximum."""
    from morphganformer_amd.iresnet import BiometricLoss, IResNetEmbedder, random_state
    from oracle.embed_ref import biometric_loss_ref
    torch.manual_seed(size)
    n = 2
    sd_np = random_state(18, seed=3)
    pred = (torch.rand(n, 3, size, size, dtype=torch.float64) * 2 - 1).requires_grad_(True)
    target = torch.rand(1, 3, size, size, dtype=torch.float64) * 2 - 1
    val = biometric_loss_ref({k: torch.from_numpy(v).double() for k, v in sd_np.items()}, pred, target.expand(n, -1, -1, -1), 18)
    (ref,) = torch.autograd.grad(val.sum() * 0.3, pred)
    bio = BiometricLoss(IResNetEmbedder(sd_np, depth=18, n=n, device="cuda"))
    bio.set_target(target.float().cuda())
    out = torch.empty(n, device="cuda")
    bio.distance_into(out, pred.detach().float().cuda())
    assert rel(out, val) < 1e-4
    dimg = torch.full((n, 3, size, size), 0.5, device="cuda")
    bio.grad_into(dimg, scale=0.3, accumulate=True)
    first = dimg.clone()
    bio.grad_into(dimg, scale=0.3)
    assert torch.allclose(first - 0.5, dimg, rtol=0, atol=1e-5 * float(ref.abs().max()))      # accumulate adds to what was there
    err = (dimg.double().cpu() - ref).abs() / ref.abs().max()
    assert float(err.median()) < 1e-5 and float(err.square().mean().sqrt()) < 1e-3


@pytest.mark.parametrize("net", ["vgg", "alex"])
def test_gradient_projection_with_vgg_and_biometric_terms(tiny, net):
    """The full north-star objective in gradient mode (LPIPS(vgg | alex) + lamda Wing + beta MSE + gamma embedding MSE): runs as a
    replayed graph, moves the latent, and its first loss equals the literal engine's loss of the same candidate."""
    from morphganformer_amd.iresnet import BiometricLoss, IResNetEmbedder
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs, ProjectionEngine, synthetic_landmarks
    gg, tsd, cfg = tiny
    G = gg.G
    steps = 6
    torch.manual_seed(2)
    latent_mean = torch.randn(cfg.k, cfg.z_dim, device="cuda")
    eps = torch.randn(steps, 1, cfg.k, cfg.z_dim, device="cuda")
    target = G(torch.randn(1, cfg.k, cfg.z_dim, device="cuda"), None, noise_mode="const")[0].clamp(-1, 1).clone()
    lm_t, lm_s = synthetic_landmarks(steps, 64, 9)
    args = ProjectionArgs(step=steps, lr=0.05, lr_rampup=0.2, min_loss_init=1e30)
    mk = lambda cls, **kw: cls(G, target, latent_mean, 1.0, args, percept=PerceptualLoss(net=net, allow_random_backbone=True), lm_target=lm_t, lm_steps=lm_s,
                               eps=eps, noise_mode="const", biometric=BiometricLoss(IResNetEmbedder(None, depth=18, n=1, device="cuda")),
                               gamma=1e-3, **kw)
    lit = mk(ProjectionEngine, batch=1, use_graph=False).run(1)
    first_literal = float(lit.losses[0])
    eng = mk(GradientProjectionEngine, use_graph=True).run()
    lat, bstep, bloss, losses = eng.result()
    assert np.isfinite(losses).all()
    assert abs(losses[0] - first_literal) < 1e-4 * abs(first_literal)       # step 0: lr = 0, same candidate, same objective
    assert float((eng.latent_in[0] - latent_mean).abs().max()) > 0.01


def test_gradient_projection_lockstep_targets_equal_single_runs(tiny):
    """B = 2 targets in one engine (one generator forward/backward per step for both) against two single-target engines on the
    same noise streams: same losses, same latent trajectories, same best steps; a "no face" step of one target leaves the other
    untouched."""
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs, synthetic_landmarks
    gg, tsd, cfg = tiny
    G = gg.G
    steps, B = 8, 2
    torch.manual_seed(21)
    latent_mean = torch.randn(cfg.k, cfg.z_dim, device="cuda")
    eps = torch.randn(steps, B, cfg.k, cfg.z_dim, device="cuda")
    targets = G(torch.randn(B, cfg.k, cfg.z_dim, device="cuda"), None, noise_mode="const")[0].clamp(-1, 1).clone()
    lms = [synthetic_landmarks(steps, 64, 9 + j) for j in range(B)]
    valid = np.ones((B, steps), np.int32)
    valid[1, 2] = 0
    args = ProjectionArgs(step=steps, lr=0.05, lr_rampup=0.25)
    singles = []
    for j in range(B):
        e = GradientProjectionEngine(G, targets[j:j + 1].contiguous(), latent_mean, 1.0, args, percept=PerceptualLoss(net="squeeze", allow_random_backbone=True),
                                     lm_target=lms[j][0], lm_steps=lms[j][1], lm_valid=valid[j], eps=eps[:, j:j + 1].contiguous(),
                                     noise_mode="const", use_graph=False).run()
        singles.append((e.result(), e.latent_in.cpu().clone()))
    multi = GradientProjectionEngine(G, targets, latent_mean, 1.0, args, percept=PerceptualLoss(net="squeeze", allow_random_backbone=True),
                                     lm_target=np.stack([l[0] for l in lms]), lm_steps=np.stack([l[1] for l in lms]), lm_valid=valid,
                                     eps=eps, noise_mode="const", use_graph=True).run()
    lat, bstep, bloss, losses = multi.result()
    for j in range(B):
        (slat, sstep, sloss, slosses), sfinal = singles[j]
        ok = ~np.isnan(slosses)
        assert np.array_equal(np.isnan(losses[j]), np.isnan(slosses))
        # tight on the first two steps, loose afterwards: Adam's first updates are sign-like (g / sqrt(v) with a young v), so the two dispatches' 1e-6 rounding
        # differences grow 5 - 10 x per step at lr 0.05 (suite soaks drew 1e-6, 3e-6, 2e-5, 1e-4, 1e-3 over steps 0 .. 4: tools/soak_lockstep_probe.py)
        first = ok & (np.arange(steps) < 2)
        assert np.abs(losses[j][first] - slosses[first]).max() < 1e-5 * np.abs(slosses[ok]).max()
        assert np.abs(losses[j][ok] - slosses[ok]).max() < 5e-2 * np.abs(slosses[ok]).max()
        assert int(bstep[j]) == sstep
        assert float((multi.latent_in[j].cpu() - sfinal[0]).abs().max()) < 0.25 * args.lr * steps      # (one sign-like Adam update of a coordinate is lr: see above)
    assert np.isnan(losses[1, 2]) and not np.isnan(losses[0, 2])


def test_lockstep_targets_with_the_biometric_term(tiny):
    """B = 2 targets with the full objective (LPIPS + Wing + MSE + embedding MSE, one target embedding each): first-step losses equal
    those of two single-target engines."""
    from morphganformer_amd.iresnet import BiometricLoss, IResNetEmbedder
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs, synthetic_landmarks
    gg, tsd, cfg = tiny
    G = gg.G
    steps, B = 4, 2
    torch.manual_seed(31)
    latent_mean = torch.randn(cfg.k, cfg.z_dim, device="cuda")
    eps = torch.randn(steps, B, cfg.k, cfg.z_dim, device="cuda")
    targets = G(torch.randn(B, cfg.k, cfg.z_dim, device="cuda"), None, noise_mode="const")[0].clamp(-1, 1).clone()
    lms = [synthetic_landmarks(steps, 64, 9 + j) for j in range(B)]
    args = ProjectionArgs(step=steps, lr=0.05, lr_rampup=0.25, min_loss_init=1e30)
    firsts = []
    for j in range(B):
        e = GradientProjectionEngine(G, targets[j:j + 1].contiguous(), latent_mean, 1.0, args, percept=PerceptualLoss(net="squeeze", allow_random_backbone=True),
                                     lm_target=lms[j][0], lm_steps=lms[j][1], eps=eps[:, j:j + 1].contiguous(), noise_mode="const", use_graph=False,
                                     biometric=BiometricLoss(IResNetEmbedder(None, depth=18, n=1, device="cuda")), gamma=1e-3).run(2)
        firsts.append(e.losses.cpu().numpy()[:2])
    multi = GradientProjectionEngine(G, targets, latent_mean, 1.0, args, percept=PerceptualLoss(net="squeeze", allow_random_backbone=True),
                                     lm_target=np.stack([l[0] for l in lms]), lm_steps=np.stack([l[1] for l in lms]), eps=eps,
                                     noise_mode="const", use_graph=True,
                                     biometric=BiometricLoss(IResNetEmbedder(None, depth=18, n=B, device="cuda")), gamma=1e-3).run(2)
    got = multi.losses.cpu().numpy()
    for j in range(B):
        assert np.abs(got[j, :2] - firsts[j]).max() < 1e-3 * np.abs(firsts[j]).max()
