"""GPU parity of the operator-level HIP kernels (through the C ABI) against the reference's own outputs
(tests/golden/ops_*.npz, generated from /root/reference by oracle/make_golden.py) and against the CPU oracle on
seeded inputs.  Tolerances: float32 element-wise ops 1e-5 relative to max|y|; MFMA convs 1e-4 (re-associated f32 sums).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.asarray(a)).cuda()


def rel_err(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_library_loads_and_device_ok():
    from morphganformer_amd import _lib
    L = _lib.lib()
    assert L.mgf_version() >= 100
    assert L.mgf_device_ok() == 1, L.mgf_last_error()


@pytest.mark.parametrize("clamp", [None, 0.5])
def test_bias_act_golden_fwd_grad(golden, clamp):
    from morphganformer_amd.torch_utils.ops import bias_act
    from oracle.ops_ref import ACT_NAMES
    g = golden("ops_bias_act.npz")
    for act in ACT_NAMES:
        tag = f"{act}_c{'none' if clamp is None else clamp}"
        x = dev(g["x"]).requires_grad_(True)
        b = dev(g["b"])
        y = bias_act.bias_act(x, b, dim=1, act=act, clamp=clamp)
        assert rel_err(y, g[f"y_{tag}"]) < 2e-6, (act, "fwd")
        (dx,) = torch.autograd.grad(y, x, dev(g["dy"]), create_graph=True)
        assert rel_err(dx, g[f"dx_{tag}"]) < 5e-6, (act, "grad1")
        if dx.requires_grad and np.abs(g[f"d2_{tag}"]).max() > 0:
            (d2,) = torch.autograd.grad(dx, x, dev(g["ddx"]), allow_unused=True)
            assert d2 is not None and rel_err(d2, g[f"d2_{tag}"]) < 2e-5, (act, "grad2")


def test_bias_act_dim0_alpha_gain_and_dtypes(golden):
    from morphganformer_amd.torch_utils.ops import bias_act
    from oracle.ops_ref import bias_act_ref
    g = golden("ops_bias_act.npz")
    y = bias_act.bias_act(dev(g["x2"]), dev(g["b2"]), dim=0, act="lrelu", alpha=0.3, gain=1.7)
    assert rel_err(y, g["y2"]) < 2e-6
    torch.manual_seed(3)
    # float64 runs in double but alpha/gain/clamp cross the ABI as float, exactly like the plugin (bias_act.cpp:24)
    for dt, tol in ((torch.float64, 1e-7), (torch.float16, 2e-3)):
        x = torch.randn(3, 8, 5, 6, dtype=torch.float64).to(dt)
        b = torch.randn(8, dtype=torch.float64).to(dt)
        for act in ("lrelu", "swish", "tanh", "softplus", "sigmoid", "elu"):
            ref = bias_act_ref(x.double(), b.double(), act=act)
            out = bias_act.bias_act(x.cuda(), b.cuda(), act=act)
            # (activations without a float alpha / gain are double all the way: a suite soak found the float64 path converting its INPUTS to float)
            tol_a = 1e-14 if (dt == torch.float64 and act in ("tanh", "softplus", "sigmoid", "elu")) else tol
            assert out.dtype == dt and rel_err(out, ref) < tol_a, (dt, act)
    # channels_last input keeps its layout and values
    x = torch.randn(2, 6, 5, 7).cuda().to(memory_format=torch.channels_last)
    b = torch.randn(6).cuda()
    out = bias_act.bias_act(x, b, act="lrelu")
    assert out.is_contiguous(memory_format=torch.channels_last)
    assert rel_err(out, bias_act_ref(x.cpu(), b.cpu(), act="lrelu")) < 2e-6
    # empty tensor and odd sizes (scalar tail path)
    assert bias_act.bias_act(torch.empty(0, 4).cuda(), None).numel() == 0
    x = torch.randn(7, 3, 5).cuda()
    assert rel_err(bias_act.bias_act(x, torch.ones(3).cuda(), act="relu"), bias_act_ref(x.cpu(), torch.ones(3), act="relu")) < 2e-6


def test_bias_act_small_arguments_and_the_kink_at_zero():
    """Per-ELEMENT relative accuracy where the exponential formulas cancel (found by tools/fuzz_soak.sh on a one-element tensor: the plugin's
    (e^x - e^-x) / (e^x + e^-x) is 2e-5 off at |x| ~ 1e-2 in float32) and the gradient at exactly zero (relu: 0, like the reference and its plugin)."""
    from morphganformer_amd.torch_utils.ops import bias_act
    x = torch.cat([torch.logspace(-6, 0, 61), -torch.logspace(-6, 0, 61)]).float()
    for act, fn in (("tanh", torch.tanh), ("elu", torch.nn.functional.elu), ("selu", torch.nn.functional.selu), ("softplus", torch.nn.functional.softplus),
                    ("sigmoid", torch.sigmoid)):
        y = bias_act.bias_act(x.cuda(), None, act=act).cpu().double()
        want = fn(x.double())
        assert float(((y - want).abs() / want.abs().clamp_min(1e-30)).max()) < 1e-6, act
    xz = torch.tensor([[0.0, -1.0, 2.0, 0.0]], device="cuda", requires_grad=True)
    (g,) = torch.autograd.grad(bias_act.bias_act(xz, torch.zeros(4, device="cuda"), act="relu").sum(), xz)
    assert g.cpu().tolist() == [[0.0, 0.0, float(np.float32(np.sqrt(2))), 0.0]]


def test_bias_act_errors():
    from morphganformer_amd import _lib
    from morphganformer_amd.torch_utils.ops import bias_act
    with pytest.raises(_lib.MgfError):
        bias_act.bias_act(torch.zeros(2, 3), None)                    # CPU tensor: no fallback
    with pytest.raises(_lib.MgfError):
        bias_act.bias_act(torch.zeros(2, 3).cuda(), torch.zeros(4).cuda())
    with pytest.raises(NotImplementedError):
        bias_act.bias_act(torch.zeros(2, 3).cuda(), None, impl="ref")


def test_upfirdn2d_golden(golden):
    from morphganformer_amd.torch_utils.ops import upfirdn2d
    from oracle.make_golden import UPFIRDN_CASES
    g = golden("ops_upfirdn2d.npz")
    for name, shape, taps, up, down, pad, gain, flip in UPFIRDN_CASES:
        y = upfirdn2d.upfirdn2d(dev(g[f"x_{name}"]), dev(g[f"f_{name}"]), up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
        assert tuple(y.shape) == g[f"y_{name}"].shape, name
        assert rel_err(y, g[f"y_{name}"]) < 3e-6, name


@pytest.mark.parametrize("up,pad,shape", [(1, [1, 1, 1, 1], (2, 5, 65, 65)), (2, [2, 1, 2, 1], (1, 3, 40, 40)),
                                           (1, [1, 1, 1, 1], (1, 2, 129, 200)), (2, [2, 1, 2, 1], (2, 2, 33, 70))])
def test_upfirdn2d_tiled_vs_oracle(up, pad, shape):
    """The LDS-tiled fast path (>=32 wide, 4x4, up 1/2) incl. ragged tile edges, channels_last and setup_filter."""
    from morphganformer_amd.torch_utils.ops import upfirdn2d
    from oracle.ops_ref import setup_filter_ref, upfirdn2d_ref
    torch.manual_seed(5)
    x = torch.randn(*shape)
    f = setup_filter_ref([1, 3, 3, 1])
    assert torch.equal(upfirdn2d.setup_filter([1, 3, 3, 1]), f)
    ref = upfirdn2d_ref(x, f, up=up, padding=pad, gain=4.0)
    out = upfirdn2d.upfirdn2d(x.cuda(), f.cuda(), up=up, padding=pad, gain=4.0)
    assert rel_err(out, ref) < 3e-6
    out_cl = upfirdn2d.upfirdn2d(x.cuda().to(memory_format=torch.channels_last), f.cuda(), up=up, padding=pad, gain=4.0)
    assert rel_err(out_cl, ref) < 3e-6
    # gradient = self-application with swapped factors
    xg = x.cuda().requires_grad_(True)
    (gx,) = torch.autograd.grad(upfirdn2d.upfirdn2d(xg, f.cuda(), up=up, padding=pad, gain=4.0).square().sum(), xg)
    xr = x.clone().requires_grad_(True)
    (gr,) = torch.autograd.grad(upfirdn2d_ref(xr, f, up=up, padding=pad, gain=4.0).square().sum(), xr)
    assert rel_err(gx, gr) < 1e-5


@pytest.mark.parametrize("shape", [(2, 3, 64, 64), (1, 2, 128, 200), (1, 1, 36, 40), (2, 2, 200, 132)])
def test_blur_gradient_geometry_on_the_separable_kernel(shape):
    """The gradient of the post-transposed-conv blur is the same filter with pad 2 on every side and an output one larger than the input
    (odd width: rows are not 16-byte aligned) -- the separable kernel's PADX = 2 / ragged instantiation, vs the oracle."""
    from morphganformer_amd import conv as cv
    from oracle.ops_ref import setup_filter_ref, upfirdn2d_ref
    torch.manual_seed(shape[2] + shape[3])
    f = setup_filter_ref([1, 3, 3, 1])
    x = torch.randn(*shape)
    ref = upfirdn2d_ref(x, f, padding=[2, 2, 2, 2], gain=4.0, flip_filter=True)
    out = torch.full([shape[0], shape[1], shape[2] + 1, shape[3] + 1], 7.0, device="cuda")
    cv.upfirdn_into(out, x.cuda(), f.cuda(), up=1, pad=(2, 2, 2, 2), gain=4.0, flip=True, separable=True)
    assert tuple(out.shape) == tuple(ref.shape)
    assert rel_err(out, ref) < 3e-6


@pytest.mark.parametrize("n,c,res", [(1, 3, 64), (2, 5, 128), (1, 2, 192)])
def test_upfirdn2d_wide_kernels_vs_oracle(n, c, res):
    """The float4 kernels of the synthesis hot path: (a) blur of a padded-pitch [.., 2h+1, 2w+1] transposed-conv workspace
    with the fused noise/bias/lrelu/residual epilogue, (b) the x2 skip-path upsample; ragged bottom tiles included."""
    from morphganformer_amd import _lib, conv as cv
    from oracle.ops_ref import bias_act_ref, setup_filter_ref, upfirdn2d_ref
    torch.manual_seed(res + c)
    f = setup_filter_ref([1, 3, 3, 1])
    h = res // 2
    pitch = cv.tconv_pitch(h)
    t_full = torch.randn(n, c, 2 * h + 1, pitch)                       # pad columns hold garbage on purpose
    t = t_full[:, :, :, :2 * h + 1]
    noise, bias, resid = torch.randn(n, res, res), torch.randn(c), torch.randn(n, c, res, res)
    strength = torch.tensor([0.37])
    ref = upfirdn2d_ref(t.contiguous(), f, padding=[1, 1, 1, 1], gain=4.0)
    ref = bias_act_ref(ref + noise[:, None] * strength, bias, act="lrelu", gain=1.3) + resid
    td = t_full.cuda()[:, :, :, :2 * h + 1]
    nd, bd, rd, sd_ = noise.cuda(), bias.cuda(), resid.cuda(), strength.cuda()
    ep = _lib.make_epilogue(bias=bd, noise=nd, noise_strength=sd_, noise_n=n, act="lrelu", gain=1.3, residual=rd)
    out = torch.empty(n, c, res, res, device="cuda")
    cv.upfirdn_into(out, td, f.cuda(), up=1, pad=(1, 1, 1, 1), gain=4.0, epilogue=ep)
    assert rel_err(out, ref) < 3e-6
    out2 = torch.empty(n, c, res, res, device="cuda")
    cv.upfirdn_into(out2, td, f.cuda(), up=1, pad=(1, 1, 1, 1), gain=4.0)
    assert rel_err(out2, upfirdn2d_ref(t.contiguous(), f, padding=[1, 1, 1, 1], gain=4.0)) < 3e-6
    # (c) the separable 4x4-patch form of the blur (MGF_FILTER_SEPARABLE hint), with and without epilogue, an asymmetric 1-D
    #     tap list so that the fx / fy roles and the flip are really exercised
    fa = setup_filter_ref([1, 2, 5, 3])
    for flip in (False, True):
        refa = upfirdn2d_ref(t.contiguous(), fa, padding=[1, 1, 1, 1], gain=4.0, flip_filter=flip)
        o = torch.empty(n, c, res, res, device="cuda")
        cv.upfirdn_into(o, td, fa.cuda(), up=1, pad=(1, 1, 1, 1), gain=4.0, flip=flip, separable=True)
        assert rel_err(o, refa) < 3e-6
        o2 = torch.empty(n, c, res, res, device="cuda")
        cv.upfirdn_into(o2, td, fa.cuda(), up=1, pad=(1, 1, 1, 1), gain=4.0, flip=flip, separable=True, epilogue=ep)
        assert rel_err(o2, bias_act_ref(refa + noise[:, None] * strength, bias, act="lrelu", gain=1.3) + resid) < 3e-6
    x = torch.randn(n, c, h, h)
    out3 = torch.empty(n, c, res, res, device="cuda")
    cv.upfirdn_into(out3, x.cuda(), f.cuda(), up=2, pad=(2, 1, 2, 1), gain=4.0)
    assert rel_err(out3, upfirdn2d_ref(x, f, up=2, padding=[2, 1, 2, 1], gain=4.0)) < 3e-6


def test_upfirdn2d_helpers_vs_oracle():
    from morphganformer_amd.torch_utils.ops import upfirdn2d
    from oracle.ops_ref import setup_filter_ref, upfirdn2d_ref
    torch.manual_seed(6)
    x = torch.randn(1, 3, 12, 12)
    f = setup_filter_ref([1, 3, 3, 1])
    up = upfirdn2d.upsample2d(x.cuda(), f.cuda())
    assert rel_err(up, upfirdn2d_ref(x, f, up=2, padding=[2, 1, 2, 1], gain=4.0)) < 3e-6
    dn = upfirdn2d.downsample2d(x.cuda(), f.cuda())
    assert rel_err(dn, upfirdn2d_ref(x, f, down=2, padding=[1, 1, 1, 1])) < 3e-6
    fl = upfirdn2d.filter2d(x.cuda(), f.cuda())
    assert rel_err(fl, upfirdn2d_ref(x, f, padding=[2, 1, 2, 1])) < 3e-6
    # nearest-neighbour x4 with a separable 4-tap box (the list2tensor pattern, networks.py:1235-1237)
    box = upfirdn2d.setup_filter([1] * 8)
    assert box.ndim == 1
    big = upfirdn2d.upsample2d(x.cuda(), box.cuda(), up=8)
    ref = upfirdn2d_ref(x, setup_filter_ref([1] * 8), up=8, padding=[7, 0, 7, 0], gain=64.0)
    assert rel_err(big, ref) < 3e-6
    with pytest.raises(Exception):
        upfirdn2d.upfirdn2d(torch.zeros(1, 1, 2, 2).cuda(), torch.ones(5, 5).cuda())      # empty output


def test_modulated_conv_golden(golden):
    from morphganformer_amd.torch_utils.ops import conv2d_resample
    g = golden("ops_modconv.npz")
    x, w, s, f = dev(g["x"]), dev(g["w"]), dev(g["s"]), dev(g["f"])
    for up in (1, 2):
        for demod in (True, False):
            y = conv2d_resample.modulated_conv2d(x, w, s, up=up, padding=1, resample_kernel=f, demodulate=demod,
                                                 flip_weight=(up == 1))
            ref = g[f"y_up{up}_demod{int(demod)}"]
            assert tuple(y.shape) == ref.shape
            assert rel_err(y, ref) < 1e-5, (up, demod)
    y = conv2d_resample.conv2d_resample(x, dev(g["w1"]), f=f, up=2, padding=0, flip_weight=False)
    assert rel_err(y, g["y_skip_up2"]) < 1e-5


@pytest.mark.parametrize("up,down,k,flip_w", [(1, 1, 3, True), (1, 1, 3, False), (1, 1, 1, True), (2, 1, 3, False), (2, 1, 3, True), (2, 1, 1, True),
                                               (1, 2, 3, True), (1, 2, 1, True), (2, 2, 3, True), (1, 1, 5, True)])
def test_conv2d_resample_gradient_vs_oracle_autograd(up, down, k, flip_w):
    """The operator is differentiable like the reference's (conv2d_gradfix.py:50-162 under conv2d_resample.py:51-146): first-order dL/dx against
    float64 autograd through the oracle's restatement for up / down in {1, 2}, 1x1 / 3x3 (+ a 25-tap kernel on the chained launches), and
    the second-order piece a gradient penalty needs -- d<dx, v>/d(dy) = the forward applied to v."""
    from morphganformer_amd.torch_utils.ops import conv2d_resample as cr
    from oracle.ops_ref import conv2d_resample_ref, setup_filter_ref
    torch.manual_seed(up * 100 + down * 10 + k)
    n, ci, co, h = 2, 12, 20, 14
    x = torch.randn(n, ci, h, h)
    w = torch.randn(co, ci, k, k) / math.sqrt(ci * k * k)
    f = setup_filter_ref([1, 3, 3, 1]) if (up > 1 or down > 1) else None
    pad = k // 2
    xr = x.double().requires_grad_(True)
    ref = conv2d_resample_ref(xr, w.double(), None if f is None else f.double(), up=up, down=down, padding=pad, flip_weight=flip_w)
    gy = torch.randn(ref.shape, dtype=torch.float64)
    (dx_ref,) = torch.autograd.grad(ref, xr, gy)
    xd = x.cuda().requires_grad_(True)
    y = cr.conv2d_resample(xd, w.cuda(), None if f is None else f.cuda(), up=up, down=down, padding=pad, flip_weight=flip_w)
    assert y.requires_grad and y.grad_fn is not None
    assert rel_err(y, ref) < 2e-5
    gyd = gy.float().cuda().requires_grad_(True)
    (dx,) = torch.autograd.grad(y, xd, gyd, create_graph=True)
    assert tuple(dx.shape) == tuple(x.shape) and rel_err(dx, dx_ref) < 2e-5, (up, down, k)
    v = torch.randn(x.shape)
    (d2,) = torch.autograd.grad((dx * v.cuda()).sum(), gyd)
    ref_v = conv2d_resample_ref(v.double(), w.double(), None if f is None else f.double(), up=up, down=down, padding=pad, flip_weight=flip_w)
    assert rel_err(d2, ref_v) < 2e-5, (up, down, k)


@pytest.mark.parametrize("up,demod", [(1, True), (1, False), (2, True), (2, False)])
def test_modulated_conv2d_gradients_vs_oracle_autograd(up, demod):
    """networks.py:253-328 differentiated with respect to x AND the styles (through the modulation and the demodulation, :288-291) against float64
    autograd through the oracle's grouped-conv restatement; with noise, whose add must not cut the graph."""
    from morphganformer_amd.torch_utils.ops import conv2d_resample as cr
    from oracle.ops_ref import modulated_conv2d_ref, setup_filter_ref
    torch.manual_seed(7 + up + int(demod))
    n, ci, co, h = 3, 16, 24, 10
    x, s = torch.randn(n, ci, h, h), torch.randn(n, ci) + 1.0
    w = torch.randn(co, ci, 3, 3)
    f = setup_filter_ref([1, 3, 3, 1])
    noise = torch.randn(n, 1, h * up, h * up)
    xr, sr = x.double().requires_grad_(True), s.double().requires_grad_(True)
    ref = modulated_conv2d_ref(xr, w.double(), sr, noise.double(), up=up, padding=1, resample_kernel=f.double(), demodulate=demod, flip_weight=(up == 1))
    gy = torch.randn(ref.shape, dtype=torch.float64)
    dx_ref, ds_ref = torch.autograd.grad(ref, (xr, sr), gy)
    xd, sd = x.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    y = cr.modulated_conv2d(xd, w.cuda(), sd, noise.cuda(), up=up, padding=1, resample_kernel=f.cuda(), demodulate=demod, flip_weight=(up == 1))
    assert rel_err(y, ref) < 2e-5
    dx, ds = torch.autograd.grad(y, (xd, sd), gy.float().cuda())
    assert rel_err(dx, dx_ref) < 5e-5 and rel_err(ds, ds_ref) < 5e-5, (up, demod)
    # without a graph the noise add is the library's own pass (upfirdn2d's epilogue port), same numbers
    with torch.no_grad():
        y2 = cr.modulated_conv2d(x.cuda(), w.cuda(), s.cuda(), noise.cuda(), up=up, padding=1, resample_kernel=f.cuda(), demodulate=demod,
                                 flip_weight=(up == 1))
    assert not y2.requires_grad and rel_err(y2, ref) < 2e-5


def test_conv2d_resample_refuses_what_it_cannot_differentiate():
    """Never a silently detached tensor: trainable weights, groups > 1 and second-order style gradients raise."""
    from morphganformer_amd import _lib
    from morphganformer_amd.torch_utils.ops import conv2d_resample as cr
    x = torch.randn(1, 4, 8, 8).cuda().requires_grad_(True)
    w = torch.randn(6, 4, 3, 3).cuda()
    with pytest.raises(_lib.MgfError, match="constants"):
        cr.conv2d_resample(x, w.clone().requires_grad_(True), padding=1)
    with pytest.raises(_lib.MgfError, match="constants"):
        cr.modulated_conv2d(x, w.clone().requires_grad_(True), torch.ones(1, 4).cuda(), padding=1)
    with pytest.raises(_lib.MgfError, match="groups"):
        cr.conv2d_resample(x, w[:, :2].contiguous(), padding=1, groups=2)
    with torch.no_grad():                                          # inference with a trainable module's weights is fine
        cr.conv2d_resample(x, w.clone().requires_grad_(True), padding=1)
    s = torch.ones(1, 4).cuda().requires_grad_(True)
    y = cr.modulated_conv2d(x, w, s, padding=1)
    (ds,) = torch.autograd.grad(y.sum(), s, create_graph=False)
    assert ds.shape == s.shape
    y = cr.modulated_conv2d(x, w, s, padding=1)
    with pytest.raises(_lib.MgfError, match="second-order"):
        torch.autograd.grad(y.sum(), s, create_graph=True)


def test_fma_matches_reference_semantics_and_gradients():
    """fma(a, b, c) = a * b + c with broadcasting and the reference's custom backward (fma.py:12-37: products un-broadcast to each operand)."""
    from morphganformer_amd.torch_utils.ops import fma
    torch.manual_seed(3)
    a, b, c = torch.randn(2, 5, 7, 3), torch.randn(5, 1, 3), torch.randn(1, 5, 1, 1)
    ar, br, cr_ = (t.double().requires_grad_(True) for t in (a, b, c))
    ref = ar * br + cr_
    g = torch.randn(ref.shape, dtype=torch.float64)
    refs = torch.autograd.grad(ref, (ar, br, cr_), g)
    ad, bd, cd = (t.cuda().requires_grad_(True) for t in (a, b, c))
    y = fma.fma(ad, bd, cd)
    assert rel_err(y, ref) < 1e-6
    outs = torch.autograd.grad(y, (ad, bd, cd), g.float().cuda())
    for o, r, t in zip(outs, refs, (a, b, c)):
        assert tuple(o.shape) == tuple(t.shape) and rel_err(o, r) < 1e-5


@pytest.mark.parametrize("n,cin,cout,res,k,stride,pad", [
    (1, 32, 32, 64, 3, 1, 1), (2, 64, 96, 40, 3, 1, 1), (1, 3, 64, 67, 3, 2, 0), (1, 16, 64, 31, 1, 1, 0),
    (1, 48, 192, 17, 3, 1, 1), (3, 8, 3, 36, 1, 1, 0), (1, 128, 64, 8, 3, 1, 1), (1, 512, 512, 4, 3, 1, 1)])
def test_conv_taps_vs_torch(n, cin, cout, res, k, stride, pad):
    """FP32-MFMA tap conv vs torch CPU conv2d on seeded inputs: ragged tiles, channel padding, stride 2, 1x1, tiny maps."""
    from morphganformer_amd import conv as cv
    torch.manual_seed(n * 1000 + cin + cout + res)
    x = torch.randn(n, cin, res, res)
    w = torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)
    b = torch.randn(cout)
    ref = torch.relu(torch.nn.functional.conv2d(x, w, b, stride=stride, padding=pad))
    from morphganformer_amd import _lib
    pc = cv.pack_weights(w.cuda())
    b_dev = b.cuda()
    ep = _lib.make_epilogue(bias=b_dev, act="relu")
    out = cv.conv_forward(x.cuda(), pc, stride=stride, pad=(pad, pad), epilogue=ep)
    assert tuple(out.shape) == tuple(ref.shape)
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("n,cin,cout,h,w,ctotal,choff,act,res", [
    (2, 64, 16, 31, 31, 16, 0, "relu", False),        # Fire squeeze: odd map, 16 of 32 padded channels
    (3, 16, 64, 63, 63, 128, 64, "relu", False),      # Fire expand1x1 into the second half of the concat buffer
    (1, 48, 192, 15, 17, 384, 0, "relu", False),      # 3 wave columns (one idle wave)
    (2, 512, 512, 32, 32, 512, 0, "linear", False),   # resnet skip 512 -> 512: two channel tiles, XCD order, 16-byte path
    (2, 128, 64, 64, 64, 64, 0, "lrelu", True),       # residual, aligned
    (1, 6, 40, 9, 13, 50, 7, "linear", True),         # ragged everything: cin not a multiple of 8, residual on an odd slice
    (2, 7, 40, 9, 13, 40, 0, "lrelu", False),         # ODD cin: the last k-step has one live channel (descriptor of one plane / weight row / style value)
    (1, 31, 64, 12, 12, 64, 0, "relu", True),         # ... on the two-blocks-per-wave shape (K <= 32, 64 output channels)
    (1, 33, 96, 5, 7, 96, 0, "linear", False),        # ... past one load group
    (1, 2, 3, 1, 1, 3, 0, "linear", False),           # one pixel
    (2, 32, 32, 8, 8, 32, 0, "linear", False),        # split-K workgroups with fewer channel groups than waves
    (1, 72, 40, 16, 12, 40, 0, "relu", True),         # ... and with an odd number of groups per wave
    (1, 512, 512, 32, 32, 512, 0, "lrelu", False),    # ... one 32-pixel block per wave (few workgroups): the gradient mode's skip at one sample
    (1, 40, 72, 5, 9, 80, 8, "lrelu", True),          # ... its ragged second pixel tile, ragged channels, residual on a slice
    (1, 256, 128, 128, 128, 128, 0, "linear", False),
    (2, 32, 3, 64, 64, 3, 0, "linear", False),        # ToRGB outside the fused launch: the narrow-output streaming kernel, 16-byte path
    (2, 13, 3, 9, 13, 8, 2, "lrelu", True),           # ... scalar path, cin not a multiple of its 8-row sweep, residual on a slice
    (1, 64, 4, 32, 36, 4, 0, "relu", True),           # ... four outputs
    (2, 3, 32, 64, 64, 32, 0, "linear", False),       # ToRGB's data gradient: the few-inputs streaming kernel, 16-byte path
    (1, 4, 37, 9, 13, 50, 5, "lrelu", True),          # ... scalar path, four inputs, ragged outputs into a slice with a residual
    (2, 1, 8, 16, 16, 8, 0, "relu", False)])          # ... one input
def test_conv1x1_register_gemm_vs_torch_and_tap_list(n, cin, cout, h, w, ctotal, choff, act, res, monkeypatch):
    """csrc/pointwise.hip (what conv_forward runs for un-modulated 1x1 layers) vs torch CPU conv2d and vs the tap-list kernel."""
    from morphganformer_amd import _lib, conv as cv
    torch.manual_seed(cin * 7 + cout + h)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 1, 1) / math.sqrt(cin)
    b = torch.randn(cout)
    r = torch.randn(n, ctotal, h, w)
    ref = torch.nn.functional.conv2d(x, wt, b)
    ref = {"relu": torch.relu, "lrelu": lambda t: torch.nn.functional.leaky_relu(t, 0.2), "linear": lambda t: t}[act](ref) * 1.5
    if res:
        ref = ref + r[:, choff:choff + cout]
    pc = cv.pack_weights(wt.cuda())
    b_dev, r_dev = b.cuda(), r.cuda()
    outs = []
    for pointwise in (True, False):
        monkeypatch.setattr(cv, "POINTWISE", pointwise)
        ep = _lib.make_epilogue(bias=b_dev, act=act, alpha=0.2, gain=1.5, residual=r_dev if res else None)
        out = torch.full([n, ctotal, h, w], 7.0, device="cuda")
        got = cv.conv_forward(x.cuda(), pc, epilogue=ep, out=out, out_choff=choff)
        assert got is out
        assert rel_err(out[:, choff:choff + cout], ref) < 2e-5
        keep = torch.ones(ctotal, dtype=torch.bool)
        keep[choff:choff + cout] = False
        assert bool((out[:, keep.cuda()] == 7.0).all())                 # nothing outside the slice is written
        outs.append(out)
    assert rel_err(outs[0], outs[1]) < 2e-6
    # no epilogue at all
    monkeypatch.setattr(cv, "POINTWISE", True)
    assert rel_err(cv.conv_forward(x.cuda(), pc), torch.nn.functional.conv2d(x, wt)) < 2e-5
    # per-sample style on the input channels (ToRGB: modulated, not demodulated), both kernels
    sty = 1 + 0.3 * torch.randn(n, cin)
    ref_m = torch.nn.functional.conv2d(x * sty[:, :, None, None], wt, b)
    for pointwise in (True, False):
        monkeypatch.setattr(cv, "POINTWISE", pointwise)
        got = cv.conv_forward(x.cuda(), pc, in_scale=sty.cuda(), epilogue=_lib.make_epilogue(bias=b_dev))
        assert rel_err(got, ref_m) < 2e-5


@pytest.mark.parametrize("n,narrow,wide,h,w", [(2, 3, 64, 67, 131), (1, 3, 64, 8, 9), (2, 1, 20, 33, 64), (1, 4, 70, 130, 66), (1, 2, 5, 3, 3)])
def test_narrow_stride2_convs_vs_torch_and_tap_list(n, narrow, wide, h, w, monkeypatch):
    """csrc/narrow_conv.hip: the 3x3 / stride-2 conv with <= 4 INPUT channels (LPIPS stem forward) and the 3x3 / stride-2 transposed
    conv with <= 4 OUTPUT channels (its data gradient) against torch CPU and against the MFMA tap-list kernel they replace."""
    from morphganformer_amd import _lib, conv as cv
    torch.manual_seed(h * 7 + w)
    x = torch.randn(n, narrow, h, w)
    wt = torch.randn(wide, narrow, 3, 3) / 3
    b = torch.randn(wide)
    ref = torch.relu(torch.nn.functional.conv2d(x, wt, b, stride=2))
    pc = cv.pack_weights(wt.cuda())
    got = cv.conv3x3s2_few_inputs(x.cuda(), wt.cuda(), bias=b.cuda(), relu=True)
    assert tuple(got.shape) == tuple(ref.shape) and rel_err(got, ref) < 2e-5
    taps = cv.conv_forward(x.cuda(), pc, stride=2, pad=(0, 0), epilogue=_lib.make_epilogue(bias=b.cuda(), act="relu"))
    assert rel_err(got, taps) < 2e-6
    lin = cv.conv3x3s2_few_inputs(x.cuda(), wt.cuda())            # no bias, no activation
    assert rel_err(lin, torch.nn.functional.conv2d(x, wt, stride=2)) < 2e-5
    # the data gradient: dy [n, wide, oh, ow] -> dx on the (2 oh + 1) x (2 ow + 1) grid the transposed conv writes
    dy = torch.randn_like(ref)
    ref_t = torch.nn.functional.conv_transpose2d(dy, wt, stride=2)
    pt = cv.transpose_packed(pc, flip=False)
    outs = []
    for narrow_on in (True, False):
        monkeypatch.setattr(cv, "NARROW_CONV", narrow_on)
        buf = torch.full([n, narrow, 2 * ref.shape[2] + 1, cv.tconv_pitch(ref.shape[3])], 7.0, device="cuda")
        got_t = cv.tconv3x3s2_forward(dy.cuda(), pt, out=buf)
        assert tuple(got_t.shape) == tuple(ref_t.shape) and rel_err(got_t, ref_t) < 2e-5
        outs.append(got_t.clone())
    assert rel_err(outs[0], outs[1]) < 2e-6


def test_conv1x1_refuses_a_leaky_slope_outside_the_unit_interval():
    """The 1x1 epilogue forms max(t, slope t), which is leaky ReLU only for 0 <= slope <= 1: anything else is an error, not a wrong answer."""
    from morphganformer_amd import _lib, conv as cv
    x = torch.randn(1, 16, 8, 8, device="cuda")
    pc = cv.pack_weights(torch.randn(32, 16, 1, 1, device="cuda"))
    for alpha in (1.5, -0.1):
        with pytest.raises(_lib.MgfError, match="slope"):
            cv.conv_forward(x, pc, epilogue=_lib.make_epilogue(act="lrelu", alpha=alpha))
    cv.conv_forward(x, pc, epilogue=_lib.make_epilogue(act="lrelu", alpha=1.0))


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 40, 72, 20, 37), (1, 33, 130, 70, 16), (1, 5, 8, 16, 129)])
def test_tconv_non_square_and_ragged_channels_vs_torch(n, cin, cout, h, w):
    """Transposed conv on non-square maps with channel counts that fill neither the border kernel's 64-channel tile nor its 32-channel K
    chunk: the MFMA border kernel's row part and column part have different lengths, its last position group and channel tile are ragged."""
    from morphganformer_amd import conv as cv
    torch.manual_seed(cin + cout + h + w)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
    s = 1 + 0.2 * torch.randn(n, cin)
    d = 1 + 0.2 * torch.randn(n, cout)
    ref = torch.nn.functional.conv_transpose2d(x * s[:, :, None, None], wt.transpose(0, 1), stride=2) * d[:, :, None, None]
    out = cv.tconv3x3s2_forward(x.cuda(), cv.pack_weights(wt.cuda()), in_scale=s.cuda(), out_scale=d.cuda())
    assert tuple(out.shape) == tuple(ref.shape)
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("n,cin,cout,res", [(1, 32, 32, 16), (2, 64, 32, 33), (1, 8, 40, 4), (1, 128, 64, 64), (3, 48, 96, 40), (1, 6, 20, 130)])
def test_tconv_vs_torch(n, cin, cout, res):
    from morphganformer_amd import conv as cv
    torch.manual_seed(cin + cout + res)
    x = torch.randn(n, cin, res, res)
    w = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
    s = 1 + 0.2 * torch.randn(n, cin)
    d = 1 + 0.2 * torch.randn(n, cout)
    ref = torch.nn.functional.conv_transpose2d(x * s[:, :, None, None], w.transpose(0, 1), stride=2) * d[:, :, None, None]
    pc = cv.pack_weights(w.cuda())
    out = cv.tconv3x3s2_forward(x.cuda(), pc, in_scale=s.cuda(), out_scale=d.cuda())
    assert tuple(out.shape) == tuple(ref.shape)
    assert rel_err(out, ref) < 2e-5


def test_integration_md_plugin_stub_runs_as_written():
    """The ctypes binding that INTEGRATION.md tells a maintainer to add to the reference (torch_utils/ops/_mgf_plugin.py) is
    executed verbatim (only the library path is filled in) and checked against the oracle: the drop-in claim, tested."""
    import os
    import re
    from morphganformer_amd import _lib
    from oracle.ops_ref import bias_act_ref, upfirdn2d_ref
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# torch_utils/ops/_mgf_plugin\.py.*?)```", text, re.S).group(1)
    code = code.replace("/path/to/morphganformer_amd/libmgf_hip.so", _lib.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md:_mgf_plugin.py", "exec"), ns)
    torch.manual_seed(3)
    x = torch.randn(2, 5, 9, 7)
    b = torch.randn(5)
    empty = torch.empty(0)
    y = ns["bias_act_plugin"].bias_act(x.cuda(), b.cuda(), empty.cuda(), empty.cuda(), empty.cuda(), 0, 1, 3, 0.2, 2 ** 0.5, -1.0)
    want = bias_act_ref(x, b, dim=1, act="lrelu", alpha=0.2, gain=2 ** 0.5, clamp=None)
    assert float((y.cpu() - want).abs().max()) < 5e-6
    f = torch.tensor([1.0, 3.0, 3.0, 1.0])
    f2 = (f[:, None] * f[None, :]) / 64
    xin = torch.randn(1, 3, 10, 12)
    y2 = ns["upfirdn2d_plugin"].upfirdn2d(xin.cuda(), f2.cuda(), 2, 2, 1, 1, 2, 1, 2, 1, False, 4.0)
    want2 = upfirdn2d_ref(xin, f2, up=2, down=1, padding=[2, 1, 2, 1], flip_filter=False, gain=4.0)
    assert tuple(y2.shape) == tuple(want2.shape) and float((y2.cpu() - want2).abs().max()) < 5e-6


def test_blur_kernels_ragged_height():
    """Non-square map: the blur kernels tile 64 columns wide, the last tile row is partial (80 = 64 + 16 output rows)."""
    from morphganformer_amd import conv as cv
    from oracle.ops_ref import setup_filter_ref, upfirdn2d_ref
    torch.manual_seed(9)
    f = setup_filter_ref([1, 3, 3, 1])
    h, w = 40, 64
    pitch = cv.tconv_pitch(w)
    t_full = torch.randn(2, 3, 2 * h + 1, pitch)
    t = t_full[:, :, :, :2 * w + 1]
    ref = upfirdn2d_ref(t.contiguous(), f, padding=[1, 1, 1, 1], gain=4.0)
    td = t_full.cuda()[:, :, :, :2 * w + 1]
    for sep in (False, True):
        out = torch.full((2, 3, 2 * h, 2 * w), 7.0, device="cuda")
        cv.upfirdn_into(out, td, f.cuda(), up=1, pad=(1, 1, 1, 1), gain=4.0, separable=sep)
        assert rel_err(out, ref) < 3e-6, sep


@pytest.mark.parametrize("shape,pad,taps,flip", [((2, 3, 70, 130), [1, 1, 1, 1], [1, 3, 3, 1], False), ((1, 2, 129, 201), [2, 1, 0, 3], None, True),
                                                 ((1, 4, 256, 256), [1, 1, 1, 1], [1, 3, 3, 1], True),
                                                 # rows readable with 16-byte loads, the patch starting 2 / 3 / 0 columns off a 4-column boundary
                                                 ((1, 2, 132, 200), [2, 1, 0, 3], None, True), ((2, 1, 66, 260), [3, 0, 1, 2], None, False),
                                                 ((1, 3, 40, 128), [0, 2, 2, 0], [1, 3, 3, 1], False)])
def test_upfirdn2d_down2_tiled_vs_oracle(shape, pad, taps, flip):
    """The LDS-tiled decimating kernel (down 2, <= 4x4 filter, >= 32 outputs per row): the gradient of the skip branch's 2x
    upsampling in gradient mode; ragged tiles, asymmetric padding and a non-separable filter included."""
    from morphganformer_amd.torch_utils.ops import upfirdn2d
    from oracle.ops_ref import setup_filter_ref, upfirdn2d_ref
    torch.manual_seed(17)
    x = torch.randn(*shape)
    f = setup_filter_ref(taps) if taps is not None else torch.randn(3, 4)
    ref = upfirdn2d_ref(x, f, down=2, padding=pad, gain=4.0, flip_filter=flip)
    out = upfirdn2d.upfirdn2d(x.cuda(), f.cuda(), down=2, padding=pad, gain=4.0, flip_filter=flip)
    assert tuple(out.shape) == tuple(ref.shape) and out.shape[-1] >= 32
    assert rel_err(out, ref) < 3e-6


@pytest.mark.parametrize("form", [1, 2, 3, 21, 12, 11])
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 32, 32), (1, 16, 128, 24, 40), (3, 72, 64, 18, 34), (1, 512, 512, 16, 16), (2, 32, 32, 40, 64),
                                            (1, 20, 96, 6, 2)])
def test_winograd_conv_matches_direct_conv_and_oracle(n, cin, cout, h, w, form, monkeypatch, request):
    """Winograd F(2x2,3x3) kernels (form 1: one workgroup per CU, 64 channels x 16x16 outputs, 8-channel chunks; form 2: two
    workgroups per CU, 32 channels, 4-channel chunks, the position halves meeting through LDS; form 3: one row of the transformed
    patch per wave, transformed input kept in registers, 64 or 32 channels per workgroup) vs the 9-tap MFMA kernel and the CPU
    oracle: style modulation, demodulation, fused noise/bias/lrelu/residual epilogue, ragged tiles."""
    from morphganformer_amd import _lib, conv as cv
    from oracle.ops_ref import bias_act_ref
    if form == 1 and (cin % 8 or cout % 64):
        pytest.skip("form 1 takes 8-channel chunks and 64-channel tiles")
    monkeypatch.setattr(cv, "WINOGRAD_FORM", 3 if form >= 3 else 2)
    if form > 3:                                  # form 3 with a pinned workgroup shape (21 falls back to the automatic one when cout % 64)
        _lib.check(_lib.lib().mgf_winograd3_force_shape(form))
        request.addfinalizer(lambda: _lib.lib().mgf_winograd3_force_shape(0))
    torch.manual_seed(cin + h)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    s, d = 1 + 0.3 * torch.randn(n, cin), 0.5 + torch.rand(n, cout)
    noise, bias, resid = torch.randn(n, h, w), torch.randn(cout), torch.randn(n, cout, h, w)
    strength = torch.tensor([0.37])
    ref = torch.nn.functional.conv2d((x * s[:, :, None, None]).double(), wt.double(), padding=1) * d[:, :, None, None].double()
    ref_ep = bias_act_ref(ref.float() + noise[:, None] * strength, bias, act="lrelu", gain=1.3) + resid
    f = lambda t: t.cuda().contiguous()
    xd, sd, dd = f(x), f(s), f(d)
    u = cv.winograd_weights(f(wt), gain=1.0) if form == 1 else cv.winograd2_weights(f(wt), gain=1.0)
    pc = cv.pack_weights(f(wt))
    plain = cv.winograd_forward(xd, u, in_scale=sd, out_scale=dd)
    direct = cv.conv_forward(xd, pc, pad=(1, 1), in_scale=sd, out_scale=dd)
    assert rel_err(plain, ref) < 2e-5
    assert rel_err(plain, direct) < 2e-5
    nd, bd, rd, st = f(noise), f(bias), f(resid), strength.cuda()
    ep = _lib.make_epilogue(bias=bd, noise=nd, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.3, residual=rd)
    out = cv.winograd_forward(xd, u, in_scale=sd, out_scale=dd, epilogue=ep)
    assert rel_err(out, ref_ep) < 2e-5
    assert rel_err(cv.winograd_forward(xd, u), torch.nn.functional.conv2d(x.double(), wt.double(), padding=1)) < 2e-5


@pytest.mark.parametrize("form", [2, 3, 12])
@pytest.mark.parametrize("n,cin,h,w", [(2, 32, 64, 64), (1, 16, 24, 40), (3, 32, 20, 34)])
def test_winograd_fused_torgb_matches_tap_list_launch(n, cin, h, w, form, monkeypatch, request):
    """conv_last + ToRGB in one Winograd launch (form 2; form 3 in its 32x32-tile and 32x64-tile shapes) vs the tap-list kernel's fused
    projection and vs torch."""
    from morphganformer_amd import _lib, conv as cv
    monkeypatch.setattr(cv, "WINOGRAD_FORM", 3 if form >= 3 else 2)
    if form > 3:
        _lib.check(_lib.lib().mgf_winograd3_force_shape(form))
        request.addfinalizer(lambda: _lib.lib().mgf_winograd3_force_shape(0))
    torch.manual_seed(h + cin)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(32, cin, 3, 3) / (3 * cin ** 0.5)
    s, d = 1 + 0.3 * torch.randn(n, cin), 0.5 + torch.rand(n, 32)
    rgbw, rgbb = torch.randn(n, 3, 32), torch.randn(3)
    conv = torch.nn.functional.conv2d((x * s[:, :, None, None]).double(), wt.double(), padding=1) * d[:, :, None, None].double()
    ref = torch.einsum("nkc,nchw->nkhw", rgbw.double(), conv) + rgbb.double()[None, :, None, None]
    f = lambda t: t.cuda().contiguous()
    xd, sd, dd, rw, rb = f(x), f(s), f(d), f(rgbw), f(rgbb)
    out = torch.empty(n, 3, h, w, device="cuda")
    cv.winograd2_rgb_forward(xd, cv.winograd2_weights(f(wt)), rw, rb, out, in_scale=sd, out_scale=dd)
    assert rel_err(out, ref) < 2e-5
    direct = torch.empty_like(out)
    cv.conv_forward(xd, cv.pack_weights(f(wt)), pad=(1, 1), in_scale=sd, out_scale=dd, rgb=(rw, rb, direct))
    assert rel_err(out, direct) < 2e-5


@pytest.mark.parametrize("form", [2, 3])
@pytest.mark.parametrize("n,cin,cout,h,w,ctotal,choff", [(2, 16, 64, 63, 63, 128, 64), (1, 32, 32, 31, 45, 96, 32), (2, 48, 64, 40, 33, 64, 0),
                                                         (1, 16, 64, 255, 255, 128, 64)])
def test_winograd_odd_maps_and_channel_slices(n, cin, cout, h, w, ctotal, choff, form, monkeypatch):
    """Forms 2 and 3 on odd map sides writing a channel slice of a wider buffer (the Fire expand3x3 half of a concat buffer) with the
    bias+ReLU epilogue; the rest of the buffer must stay untouched."""
    from morphganformer_amd import _lib, conv as cv
    monkeypatch.setattr(cv, "WINOGRAD_FORM", form)
    torch.manual_seed(h * w)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    bias = torch.randn(cout)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1))
    f = lambda t: t.cuda().contiguous()
    out = torch.full((n, ctotal, h, w), 7.0, device="cuda")
    bd = f(bias)
    cv.winograd2_forward(f(x), cv.winograd2_weights(f(wt)), epilogue=_lib.make_epilogue(bias=bd, act="relu"), out=out, out_choff=choff)
    assert rel_err(out[:, choff:choff + cout], ref) < 2e-5
    mask = torch.ones(ctotal, dtype=torch.bool)
    mask[choff:choff + cout] = False
    assert bool((out[:, mask.cuda()] == 7.0).all())


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 32, 32, 64, 64), (1, 16, 64, 20, 36), (3, 8, 96, 8, 2), (2, 16, 32, 12, 40), (1, 8, 32, 6, 104)])
def test_winograd3_half_resolution_residual(n, cin, cout, h, w):
    """Form 3 with the resnet skip branch given at half resolution: the epilogue's in-place 2x up-sampling ([1,3,3,1] filter, padding
    [2,1,2,1], gain 4) against the reference-pinned upfirdn2d oracle, and against the two-launch path (FIR pass + full-size residual)."""
    from morphganformer_amd import _lib, conv as cv
    from oracle.ops_ref import bias_act_ref, setup_filter_ref, upfirdn2d_ref
    torch.manual_seed(cout + h)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    s, d = 1 + 0.3 * torch.randn(n, cin), 0.5 + torch.rand(n, cout)
    noise, bias, low = torch.randn(n, h, w), torch.randn(cout), torch.randn(n, cout, h // 2, w // 2)
    strength = torch.tensor([0.37])
    f = setup_filter_ref([1, 3, 3, 1])
    skip = upfirdn2d_ref(low, f, up=2, padding=[2, 1, 2, 1], gain=4.0)
    ref = torch.nn.functional.conv2d((x * s[:, :, None, None]).double(), wt.double(), padding=1) * d[:, :, None, None].double()
    ref = bias_act_ref(ref.float() + noise[:, None] * strength, bias, act="lrelu", gain=1.3) + skip
    g = lambda t: t.cuda().contiguous()
    xd, sd, dd, nd, bd, ld, st = g(x), g(s), g(d), g(noise), g(bias), g(low), strength.cuda()
    u = cv.winograd2_weights(g(wt), gain=1.0)
    ep = _lib.make_epilogue(bias=bd, noise=nd, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.3)
    out = cv.winograd_forward(xd, u, in_scale=sd, out_scale=dd, epilogue=ep, residual_low=ld)
    assert rel_err(out, ref) < 2e-5
    full = torch.empty(n, cout, h, w, device="cuda")
    cv.upfirdn_into(full, ld, g(f), up=2, pad=(2, 1, 2, 1), gain=4.0, separable=True)
    ep2 = _lib.make_epilogue(bias=bd, noise=nd, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.3, residual=full)
    two = cv.winograd_forward(xd, u, in_scale=sd, out_scale=dd, epilogue=ep2)
    assert rel_err(out, two) < 2e-6


@pytest.mark.parametrize("n,c,h,with_ep", [(8, 64, 128, True), (2, 32, 512, True), (4, 64, 256, False)])
def test_fir_streaming_kernel_vs_float64(n, c, h, with_ep):
    """fir_up1_stream (csrc/upfirdn2d.hip): the blur after the transposed conv on wide maps -- upfirdn2d(t, f, padding 1, gain 4) of the
    (2h+1)^2 padded-pitch workspace, with the fused noise / bias / lrelu epilogue -- as the no-LDS streaming kernel (256-column strips,
    DPP halo exchange, ring of four filtered rows) against float64 torch and against the LDS-tiled separable kernel it replaces (which
    serves every call below 2048 strips-x-row-segments, e.g. one channel slice of the same tensors)."""
    from morphganformer_amd import _lib, conv as cv
    torch.manual_seed(h + c)
    r = 2 * h
    f1 = torch.tensor([1., 3., 3., 1.], dtype=torch.float64)
    f2 = f1[:, None] * f1[None, :] / 64
    tbuf = torch.full((n, c, r + 1, cv.tconv_pitch(h)), float("nan"), device="cuda")       # the pad columns hold garbage in the engine
    t = tbuf[:, :, :, :r + 1]
    t.copy_(torch.randn(n, c, r + 1, r + 1, device="cuda"))
    noise, bias, st = torch.randn(n, r, r, device="cuda"), torch.randn(c, device="cuda"), torch.tensor([0.4], device="cuda")
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(t.double().reshape(n * c, 1, r + 1, r + 1), (1, 1, 1, 1)), (f2 * 4).cuda()[None, None])
    ref = ref.reshape(n, c, r, r)
    ep = None
    if with_ep:
        ep = _lib.make_epilogue(bias=bias, noise=noise, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.3)
        ref = torch.nn.functional.leaky_relu(ref + noise.double()[:, None] * 0.4 + bias.double()[None, :, None, None], 0.2) * 1.3
    y = torch.full((n, c, r, r), float("nan"), device="cuda")
    cv.upfirdn_into(y, t, f2.float().cuda(), up=1, pad=(1, 1, 1, 1), gain=4.0, epilogue=ep, separable=True)
    assert rel_err(y, ref) < 5e-6
    # one channel of one sample is far below the streaming threshold: the LDS-tiled kernel, same arithmetic
    y1 = torch.empty(1, 1, r, r, device="cuda")
    ep1 = _lib.make_epilogue(bias=bias[3:4].contiguous(), noise=noise[1:2].contiguous(), noise_strength=st, noise_n=1, act="lrelu", alpha=0.2, gain=1.3) if with_ep else None
    cv.upfirdn_into(y1, t[1:2, 3:4], f2.float().cuda(), up=1, pad=(1, 1, 1, 1), gain=4.0, epilogue=ep1, separable=True)
    assert rel_err(y[1:2, 3:4], y1) < 1e-6


@pytest.mark.parametrize("n,cout,h,w", [(2, 32, 64, 64), (1, 32, 100, 96), (3, 64, 32, 512), (1, 32, 16, 1024), (2, 32, 8, 2048),
                                        (1, 32, 160, 64), (2, 32, 136, 96)])       # 40 / 34 tile rows: not whole 32-tile columns -> the HORIZONTAL strip walk
def test_winograd3_persistent_form(n, cout, h, w):
    """wino3p_conv_kernel (the form the literal loop runs on the 1024^2 layers at 25 - 32 candidates: a workgroup walks a strip of tiles with
    its weights resident in registers, styles folded into them, the epilogue's operands by LDS-DMA) pinned through
    mgf_winograd3_force_shape(31): against float64 torch for the plain / full-resolution-residual / half-resolution-residual / no-epilogue /
    fused-ToRGB launches, and against the one-shot kernel (shape 11).  Strips of one, three and sixteen tiles, one and two strips per row,
    first / interior / last tiles of a row, top and bottom tile rows."""
    from morphganformer_amd import _lib, conv as cv
    from oracle.ops_ref import bias_act_ref, setup_filter_ref, upfirdn2d_ref
    L = _lib.lib()
    cin = 32
    torch.manual_seed(h + w + cout)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    s, d = 1 + 0.3 * torch.randn(n, cin), 0.5 + torch.rand(n, cout)
    noise, bias = torch.randn(n, h, w), torch.randn(cout)
    low, resid = torch.randn(n, cout, h // 2, w // 2), torch.randn(n, cout, h, w)
    strength = torch.tensor([0.37])
    conv = torch.nn.functional.conv2d((x * s[:, :, None, None]).double(), wt.double(), padding=1) * d[:, :, None, None].double()
    act = bias_act_ref(conv.float() + noise[:, None] * strength, bias, act="lrelu", gain=1.3)
    skip = upfirdn2d_ref(low, setup_filter_ref([1, 3, 3, 1]), up=2, padding=[2, 1, 2, 1], gain=4.0)
    g = lambda t: t.cuda().contiguous()
    xd, sd, dd, nd, bd, ld, rd, st = g(x), g(s), g(d), g(noise), g(bias), g(low), g(resid), strength.cuda()
    u = cv.winograd2_weights(g(wt), gain=1.0)
    ep = _lib.make_epilogue(bias=bd, noise=nd, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.3)
    ep_r = _lib.make_epilogue(bias=bd, noise=nd, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.3, residual=rd)
    cases = {"plain": (lambda: cv.winograd_forward(xd, u, in_scale=sd, out_scale=dd, epilogue=ep), act),
             "residual": (lambda: cv.winograd_forward(xd, u, in_scale=sd, out_scale=dd, epilogue=ep_r), act + resid),
             "half-resolution residual": (lambda: cv.winograd_forward(xd, u, in_scale=sd, out_scale=dd, epilogue=ep, residual_low=ld), act + skip),
             "no epilogue": (lambda: cv.winograd_forward(xd, u, in_scale=sd, out_scale=dd), conv.float())}
    if cout == 32:
        rw, rb = torch.randn(n, 3, cout), torch.randn(3)
        rgb_ref = torch.einsum("nco,nohw->nchw", rw.double(), conv) + rb.double()[None, :, None, None]
        cases["ToRGB"] = (lambda: cv.winograd2_rgb_forward(xd, u, g(rw), g(rb), torch.empty(n, 3, h, w, device="cuda"), in_scale=sd, out_scale=dd),
                          rgb_ref.float())
    try:
        for name, (fn, ref) in cases.items():
            _lib.check(L.mgf_winograd3_force_shape(31))
            got = fn().clone()
            _lib.check(L.mgf_winograd3_force_shape(11))
            one_shot = fn().clone()
            assert rel_err(got, ref) < 2e-5, name
            assert rel_err(got, one_shot) < 2e-6, name
    finally:
        _lib.check(L.mgf_winograd3_force_shape(0))


@pytest.mark.parametrize("n,c,f,with_ep", [(2, 256, 128, True), (1, 512, 96, True), (2, 256, 64, False), (3, 64, 80, True), (1, 512, 1024, True)])
def test_duplex_attention_forward_all_kernel_forms_vs_float64(n, c, f, with_ep):
    """mgf_duplex_attention against the folded layer written out in float64: S = x^T wqc + spos, P = softmax(S), y = x rsqrt(mean_c x^2
    + 1e-8) (P vwb^T) -> noise, bias, lrelu, gain, residual; probabilities and first-maximum assignments.  256 / 512 channels with whole
    32-pixel tiles run the MFMA kernel, the rest the register kernels."""
    import ctypes as C
    from morphganformer_amd import _lib
    L = _lib.lib()
    torch.manual_seed(c + f)
    T = 16
    x = torch.randn(n, c, f, dtype=torch.float64)
    wqc = torch.randn(c, T, dtype=torch.float64) / math.sqrt(c)
    spos = torch.randn(f, T, dtype=torch.float64)
    vwb = 1 + 0.3 * torch.randn(n, c, T, dtype=torch.float64)
    noise, bias, res = torch.randn(n, f, dtype=torch.float64), torch.randn(c, dtype=torch.float64), torch.randn(n, c, f, dtype=torch.float64)
    S = torch.einsum("ncf,ct->nft", x, wqc) + spos[None]
    P = torch.softmax(S, -1)
    r = torch.rsqrt(x.square().mean(1, keepdim=True) + 1e-8)
    y = x * r * torch.einsum("nft,nct->ncf", P, vwb)
    if with_ep:
        y = torch.nn.functional.leaky_relu(y + 0.37 * noise[:, None] + bias[None, :, None], 0.2) * 1.3 + res
    fl = lambda t: t.float().cuda().contiguous()
    xd, wd, sd_, vd, nd, bd, rd = fl(x), fl(wqc), fl(spos), fl(vwb), fl(noise), fl(bias), fl(res)
    st = torch.tensor([0.37], device="cuda")
    ep = _lib.make_epilogue(bias=bd, noise=nd, noise_strength=st, noise_n=n, act="lrelu", alpha=0.2, gain=1.3, residual=rd) if with_ep else None
    out = torch.empty_like(xd)
    probs = torch.empty(n, f, T, device="cuda")
    amax = torch.empty(n, f, dtype=torch.int32, device="cuda")
    _lib.check(L.mgf_duplex_attention(out.data_ptr(), xd.data_ptr(), wd.data_ptr(), sd_.data_ptr(), vd.data_ptr(), n, c, f, T,
                                      C.byref(ep) if ep is not None else None, 0, probs.data_ptr(), amax.data_ptr(), _lib.stream_ptr()))
    assert rel_err(out, y) < 2e-5
    assert float((probs.double().cpu() - P).abs().max()) < 1e-5
    # assignments: equal to the float64 argmax wherever the two best scores are not within rounding of each other
    top2 = S.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert bool((amax.cpu().long()[clear] == S.argmax(-1)[clear]).all())
    # without the optional outputs
    out2 = torch.empty_like(xd)
    _lib.check(L.mgf_duplex_attention(out2.data_ptr(), xd.data_ptr(), wd.data_ptr(), sd_.data_ptr(), vd.data_ptr(), n, c, f, T,
                                      C.byref(ep) if ep is not None else None, 0, None, None, _lib.stream_ptr()))
    assert torch.equal(out, out2)


def test_randn_and_rgb_weights_kernels():
    """mgf_randn_f32 (the per-layer noise maps of noise_mode="random"): Philox4x32-10 + Box-Muller with the stream position on the device --
    standard normal statistics, the same (seed, position) reproduces, consecutive launches and REPLAYS of a captured graph continue the
    stream; odd lengths and unaligned tails.  mgf_rgb_weights_f32 = W[c,co] * s[n,co] exactly."""
    from morphganformer_amd import _lib
    L, st = _lib.lib(), _lib.stream_ptr()
    n = 1 << 22
    state = torch.zeros(2, dtype=torch.int64, device="cuda")
    a, b = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    _lib.check(L.mgf_randn_f32(a.data_ptr(), n, 1234, state.data_ptr(), st))
    assert int(state[0]) == n // 4 and int(state[1]) == 0
    _lib.check(L.mgf_randn_f32(b.data_ptr(), n, 1234, state.data_ptr(), st))
    assert int(state[0]) == n // 2
    x = torch.cat([a, b]).double().cpu()
    assert abs(float(x.mean())) < 2e-3 and abs(float(x.var()) - 1) < 4e-3 and abs(float((x ** 3).mean())) < 1e-2 and abs(float((x ** 4).mean()) - 3) < 3e-2
    assert torch.isfinite(x).all() and float(x.abs().max()) > 4.5 and not torch.equal(a, b)
    assert abs(float((a[:-1] * a[1:]).mean())) < 2e-3                      # neighbours are uncorrelated
    state.zero_()
    c = torch.empty(n, device="cuda")
    _lib.check(L.mgf_randn_f32(c.data_ptr(), n, 1234, state.data_ptr(), st))
    assert torch.equal(a, c)                                               # same seed, same position: same numbers
    _lib.check(L.mgf_randn_f32(c.data_ptr(), n, 1235, state.data_ptr(), st))
    assert not torch.equal(b, c)                                           # another key
    # odd length into an unaligned tail: exactly n values written, the stream advances by ceil(n / 4)
    state.zero_()
    buf = torch.full([1003], 7.0, device="cuda")
    _lib.check(L.mgf_randn_f32(buf[1:].data_ptr(), 1001, 1234, state.data_ptr(), st))
    assert float(buf[0]) == 7.0 and float(buf[1002]) == 7.0 and int(state[0]) == 251 and torch.equal(buf[1:1001], a[:1000])
    # under graph replay every replay draws fresh numbers
    state.zero_()
    g, out = torch.cuda.CUDAGraph(), torch.empty(4096, device="cuda")
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        _lib.check(L.mgf_randn_f32(out.data_ptr(), 4096, 99, state.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        _lib.check(L.mgf_randn_f32(out.data_ptr(), 4096, 99, state.data_ptr(), _lib.stream_ptr()))
    g.replay(); first = out.clone(); g.replay(); second = out.clone()
    assert not torch.equal(first, second) and int(state[0]) == 3 * 1024
    w, s = torch.randn(3, 32), torch.randn(5, 32)
    o = torch.empty(5, 3, 32, device="cuda")
    wd, sd = w.cuda(), s.cuda()
    _lib.check(L.mgf_rgb_weights_f32(o.data_ptr(), wd.data_ptr(), sd.data_ptr(), 5, 3, 32, st))
    assert torch.equal(o.cpu(), w[None] * s[:, None])


def _build_abi_consumer(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "morphganformer_amd")
    exe = str(tmp_path / "abi_consumer")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "abi_consumer.c"),
                    "-o", exe, "-L", libdir, "-lmgf_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm"], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    return subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.gpu
def test_torch_free_c_consumer_of_the_abi(tmp_path):
    """examples/abi_consumer.c: a C99 program (gcc, no torch, no Python) linked against libmgf_hip.so and the HIP runtime calls
    mgf_bias_act / mgf_upfirdn2d / mgf_mse_f32 / mgf_dssim_u8_f32 on its own device buffers and checks them against the definitions
    restated in C -- the drop-in boundary used the way a cgo / JNI stub would use it."""
    r = _build_abi_consumer(tmp_path)
    assert r.returncode == 0 and "abi_consumer: OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
