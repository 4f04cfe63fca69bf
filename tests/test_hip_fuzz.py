"""Seeded random sweeps of the operator-level boundary (the reference's plugin / operator API: upfirdn2d, bias_act, conv2d_resample,
modulated_conv2d) against the CPU oracle: shapes, strides, paddings and filter sizes nobody picked by hand -- ragged tile edges, negative
padding (cropping), single-pixel maps, channel counts off every block size.  Each case is cheap; a failure prints its parameters.
Tolerances as in test_hip_ops.py: element-wise / FIR ops 1e-5 relative to max|y|, MFMA convolutions 2e-4 (re-associated float32 sums).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# soak runs: MGF_FUZZ_OFFSET=k shifts every seed of this file (other shapes, other data); 0 = the committed cases
_OFFSET = int(os.environ.get("MGF_FUZZ_OFFSET", "0")) * 100003


def rel_err(a, b):
    a = a.detach().double().cpu().numpy()
    b = b.detach().double().cpu().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _rng(seed):
    return np.random.default_rng(seed + _OFFSET)


def test_upfirdn2d_random_geometry():
    """up / down 1..3 per axis, 1-D and 2-D filters of 1..7 taps, per-side padding -3..5 (negative = crop), float32 / float64 / float16,
    contiguous and channels_last."""
    from morphganformer_amd.torch_utils.ops import upfirdn2d
    from oracle.ops_ref import upfirdn2d_ref
    r = _rng(101)
    done = 0
    for case in range(260):
        n, c = int(r.integers(1, 4)), int(r.integers(1, 9))
        h, w = int(r.integers(1, 48)), int(r.integers(1, 80))
        if case % 7 == 0:
            h, w = int(r.integers(60, 140)), int(r.integers(60, 140))
        upx, upy = (int(v) for v in r.integers(1, 4, 2))
        dnx, dny = (int(v) for v in r.integers(1, 4, 2))
        fh, fw = (int(v) for v in r.integers(1, 8, 2))
        one_d = bool(r.integers(0, 2))
        if one_d:
            fh = fw
        pad = [int(v) for v in r.integers(-3, 6, 4)]
        oh = (h * upy + pad[2] + pad[3] - fh + dny) // dny
        ow = (w * upx + pad[0] + pad[1] - fw + dnx) // dnx
        if h * upy + pad[2] + pad[3] < fh or w * upx + pad[0] + pad[1] < fw or oh < 1 or ow < 1:
            continue
        # (the reference crops before filtering: a crop larger than the up-sampled map is not a defined case)
        if max(-pad[2], 0) + max(-pad[3], 0) >= h * upy or max(-pad[0], 0) + max(-pad[1], 0) >= w * upx:
            continue
        flip, gain = bool(r.integers(0, 2)), float(np.float32(r.uniform(0.5, 4.0)))       # (the plugin's gain is a C float, upfirdn2d.cpp:8: a float32 value, so that the float64 gate can be tight)
        dtype = [torch.float32, torch.float32, torch.float64, torch.float16][int(r.integers(0, 4))]
        torch.manual_seed(case + _OFFSET)
        x = torch.randn(n, c, h, w)
        f = torch.rand(fw) + 0.1 if one_d else torch.rand(fh, fw) + 0.1
        if dtype == torch.float64:
            # dyadic taps and a power-of-FOUR gain: every float32 value the filter set-up may form (outer product of a 1-D filter, gain folded into the taps,
            # sqrt(gain) per pass of a separable filter -- the reference's implementations differ there, upfirdn2d.py:161-200, :222-232 / upfirdn2d.cpp:8) is
            # exact, so the float64 gate below can be tight
            f = torch.round(f * 8) / 8 + 0.125
            gain = float(4.0 ** int(r.integers(-1, 2)))
        ref = upfirdn2d_ref(x.double(), f.double(), up=(upx, upy), down=(dnx, dny), padding=pad, flip_filter=flip, gain=gain)
        xin = x.to(dtype).cuda()
        if case % 3 == 0 and c > 1:
            xin = xin.to(memory_format=torch.channels_last)
        out = upfirdn2d.upfirdn2d(xin, f.cuda(), up=[upx, upy], down=[dnx, dny], padding=pad, flip_filter=flip, gain=gain)
        tag = f"case {case}: x {tuple(x.shape)} {dtype} up ({upx},{upy}) down ({dnx},{dny}) f {'1-D ' if one_d else ''}{fh}x{fw} pad {pad} flip {flip}"
        assert tuple(out.shape) == tuple(ref.shape), tag
        tol = {torch.float16: 4e-3, torch.float32: 1e-5, torch.float64: 1e-12}[dtype]     # (float64 accumulates in double: anything above rounding is a float path)
        assert rel_err(out, ref) < tol, tag
        done += 1
    assert done > 140


def test_upfirdn2d_random_gradients():
    """The autograd of the operator is the operator applied to the gradient with up / down swapped (upfirdn2d.py:237-256): checked against
    the oracle's autograd on random geometry."""
    from morphganformer_amd.torch_utils.ops import upfirdn2d
    from oracle.ops_ref import upfirdn2d_ref
    r = _rng(202)
    done = 0
    for case in range(120):
        n, c = int(r.integers(1, 3)), int(r.integers(1, 5))
        h, w = int(r.integers(2, 40)), int(r.integers(2, 70))
        up, down = int(r.integers(1, 3)), int(r.integers(1, 3))
        ft = int(r.integers(1, 6))
        pad = [int(v) for v in r.integers(0, 4, 4)]
        if h * up + pad[2] + pad[3] < ft or w * up + pad[0] + pad[1] < ft:
            continue
        flip = bool(r.integers(0, 2))
        torch.manual_seed(1000 + case + _OFFSET)
        x = torch.randn(n, c, h, w)
        f = torch.rand(ft, ft) + 0.1
        xr = x.double().requires_grad_(True)
        yr = upfirdn2d_ref(xr, f.double(), up=up, down=down, padding=pad, flip_filter=flip, gain=2.0)
        gy = torch.randn(yr.shape, dtype=torch.float64)
        (gr,) = torch.autograd.grad((yr * gy).sum(), xr)
        xg = x.cuda().requires_grad_(True)
        yg = upfirdn2d.upfirdn2d(xg, f.cuda(), up=up, down=down, padding=pad, flip_filter=flip, gain=2.0)
        (gx,) = torch.autograd.grad((yg * gy.float().cuda()).sum(), xg)
        tag = f"case {case}: x {tuple(x.shape)} up {up} down {down} f {ft} pad {pad} flip {flip}"
        assert rel_err(gx, gr) < 1e-5, tag
        done += 1
    assert done > 80


def test_bias_act_random():
    """Every activation x dim x dtype on random ranks 1..4, with and without bias / clamp, forward and first-order gradient."""
    from morphganformer_amd.torch_utils.ops import bias_act
    from oracle.ops_ref import bias_act_ref
    acts = ["linear", "relu", "lrelu", "tanh", "sigmoid", "elu", "selu", "softplus", "swish"]
    r = _rng(303)
    for case in range(140):
        rank = int(r.integers(1, 5))
        shape = [int(v) for v in r.integers(1, 9, rank)]
        if case % 5 == 0:
            shape[-1] = int(r.integers(100, 300))
        dim = int(r.integers(0, rank))
        act = acts[int(r.integers(0, len(acts)))]
        alpha = None if r.integers(0, 2) else float(r.uniform(0.05, 0.5))
        gain = None if r.integers(0, 2) else float(r.uniform(0.5, 2.0))
        clamp = None if r.integers(0, 3) else float(r.uniform(0.2, 1.5))
        has_b = bool(r.integers(0, 2))
        dtype = [torch.float32, torch.float64, torch.float16][int(r.integers(0, 3))]
        torch.manual_seed(2000 + case + _OFFSET)
        x = torch.randn(*shape).to(dtype)                # (both sides get the values the dtype holds: a relu / selu kink between a float32 draw and
        b = torch.randn(shape[dim]).to(dtype) if has_b else None     # its float16 rounding flips that element's whole gradient)
        xr = x.double().requires_grad_(True)
        yr = bias_act_ref(xr, None if b is None else b.double(), dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
        gy = torch.randn(yr.shape, dtype=torch.float64)
        (gr,) = torch.autograd.grad((yr * gy).sum(), xr)
        xg = x.to(dtype).cuda().requires_grad_(True)
        yg = bias_act.bias_act(xg, None if b is None else b.to(dtype).cuda(), dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
        (gx,) = torch.autograd.grad((yg * gy.to(dtype).cuda()).sum(), xg)
        tag = f"case {case}: {shape} dim {dim} {act} alpha {alpha} gain {gain} clamp {clamp} bias {has_b} {dtype}"
        tol = 4e-3 if dtype == torch.float16 else 1e-5
        if dtype == torch.float64:                       # double all the way, except that alpha / gain / clamp cross the ABI as floats (bias_act.cpp:24)
            exact = gain is None and clamp is None and act in ("linear", "tanh", "sigmoid", "elu", "selu", "softplus")
            tol = 1e-13 if exact else 2e-7
        assert yg.dtype == dtype and tuple(yg.shape) == tuple(shape), tag
        assert rel_err(yg, yr) < tol, tag
        if clamp is None:                                # (at a clamp boundary the two sides may round to different sides of it)
            assert rel_err(gx, gr) < (2e-2 if dtype == torch.float16 else 1e-5), tag


def test_conv2d_resample_random():
    """1x1 / 3x3 / 5x5 kernels, up / down 1..2 with the 4-tap filter or none, padding 0..3, both weight orientations, channel counts off the
    32-wide blocks: the operator's branches (conv2d_resample.py:87-146) against the oracle's restatement."""
    from morphganformer_amd.torch_utils.ops import conv2d_resample as cr
    from oracle.ops_ref import conv2d_resample_ref, setup_filter_ref
    r = _rng(404)
    done = 0
    for case in range(160):
        n = int(r.integers(1, 4))
        cin, cout = int(r.integers(1, 80)), int(r.integers(1, 80))
        if case % 9 == 0:
            cin, cout = int(r.integers(100, 300)), int(r.integers(100, 200))
        k = [1, 3, 3, 3, 5][int(r.integers(0, 5))]
        h, w = int(r.integers(k, 40)), int(r.integers(k, 56))
        up, down = int(r.integers(1, 3)), int(r.integers(1, 3))
        if up > 1 and down > 1:
            down = 1
        if up == 2 and k == 5:
            k = 3
        use_f = (up > 1 or down > 1) or bool(r.integers(0, 2))
        f = setup_filter_ref([1, 3, 3, 1]) if use_f else None
        pad = int(r.integers(0, 4))
        flip_w = bool(r.integers(0, 2))
        torch.manual_seed(3000 + case + _OFFSET)
        x = torch.randn(n, cin, h, w)
        wt = torch.randn(cout, cin, k, k) / (k * cin ** 0.5)
        try:
            ref = conv2d_resample_ref(x.double(), wt.double(), None if f is None else f.double(), up=up, down=down, padding=pad, flip_weight=flip_w)
        except RuntimeError:
            continue                                     # (kernel larger than the padded map)
        if min(ref.shape[2:]) < 1:
            continue
        out = cr.conv2d_resample(x.cuda(), wt.cuda(), None if f is None else f.cuda(), up=up, down=down, padding=pad, flip_weight=flip_w)
        tag = f"case {case}: x {tuple(x.shape)} w {tuple(wt.shape)} up {up} down {down} pad {pad} f {use_f} flip_weight {flip_w}"
        assert tuple(out.shape) == tuple(ref.shape), tag
        assert rel_err(out, ref) < 2e-4, tag
        done += 1
    assert done > 90


def test_modulated_conv2d_random():
    """Per-sample styles, with and without demodulation / noise, up 1 and 2 (networks.py:253-328)."""
    from morphganformer_amd.torch_utils.ops import conv2d_resample as cr
    from oracle.ops_ref import modulated_conv2d_ref, setup_filter_ref
    r = _rng(505)
    f = setup_filter_ref([1, 3, 3, 1])
    for case in range(120):
        n = int(r.integers(1, 5))
        cin, cout = int(r.integers(1, 72)), int(r.integers(1, 72))
        k = [1, 3, 3][int(r.integers(0, 3))]
        up = int(r.integers(1, 3)) if k == 3 else 1
        res = int(r.integers(4, 36))
        if case % 6 == 0:
            res = int(r.integers(40, 80))
        demod, has_noise = bool(r.integers(0, 2)), bool(r.integers(0, 2))
        torch.manual_seed(4000 + case + _OFFSET)
        x = torch.randn(n, cin, res, res)
        wt = torch.randn(cout, cin, k, k)
        s = torch.randn(n, cin) + 1.0
        noise = torch.randn(n, 1, res * up, res * up) * 0.1 if has_noise else None
        ref = modulated_conv2d_ref(x.double(), wt.double(), s.double(), None if noise is None else noise.double(), up=up, padding=k // 2,
                                   resample_kernel=f.double(), demodulate=demod)
        out = cr.modulated_conv2d(x.cuda(), wt.cuda(), s.cuda(), None if noise is None else noise.cuda(), up=up, padding=k // 2,
                                  resample_kernel=f.cuda(), demodulate=demod)
        tag = f"case {case}: x {tuple(x.shape)} w {tuple(wt.shape)} up {up} demod {demod} noise {has_noise}"
        assert tuple(out.shape) == tuple(ref.shape), tag
        assert rel_err(out, ref) < 2e-4, tag


def test_conv_forward_random_epilogues():
    """The engine-level convolution entry (tap list / 1x1 GEMM / narrow streaming kernels, chosen by shape) with every epilogue piece --
    per-sample input and output scales, bias, noise, activation, gain, residual, a channel slice of a wider output buffer -- on random shapes,
    against float64 torch."""
    import torch.nn.functional as F
    from morphganformer_amd import _lib, conv as cv
    r = _rng(606)
    for case in range(120):
        n = int(r.integers(1, 4))
        cin, cout = int(r.integers(1, 100)), int(r.integers(1, 100))
        k = [1, 1, 3][int(r.integers(0, 3))]
        stride = 1 if k == 1 else int(r.integers(1, 3))
        pad = 0 if k == 1 else int(r.integers(0, 2))
        h, w = int(r.integers(k, 45)), int(r.integers(k, 61))
        act = ["linear", "relu", "lrelu"][int(r.integers(0, 3))]
        has_bias, has_noise, has_res = bool(r.integers(0, 2)), bool(r.integers(0, 2)), bool(r.integers(0, 2))
        has_is, has_os = bool(r.integers(0, 2)), bool(r.integers(0, 2))
        choff = int(r.integers(0, 9)) if r.integers(0, 2) else 0
        ctotal = cout + choff + (int(r.integers(0, 5)) if choff else 0)
        gain = float(r.uniform(0.5, 2.0))
        torch.manual_seed(5000 + case + _OFFSET)
        x = torch.randn(n, cin, h, w)
        wt = torch.randn(cout, cin, k, k) / (k * cin ** 0.5)
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        b = torch.randn(cout) if has_bias else None
        nz = torch.randn(n, oh * ow) if has_noise else None
        ns = torch.tensor([0.3])
        s_in = torch.rand(n, cin) + 0.5 if has_is else None
        s_out = torch.rand(n, cout) + 0.5 if has_os else None
        base = torch.randn(n, ctotal, oh, ow)
        ref = F.conv2d((x * s_in[:, :, None, None] if has_is else x).double(), wt.double(), stride=stride, padding=pad)
        if has_os:
            ref = ref * s_out[:, :, None, None].double()
        if has_noise:
            ref = ref + (nz.double() * 0.3).reshape(n, 1, oh, ow)
        if has_bias:
            ref = ref + b.double().reshape(1, -1, 1, 1)
        ref = {"linear": ref, "relu": ref.clamp(min=0), "lrelu": torch.where(ref > 0, ref, 0.2 * ref)}[act] * gain
        if has_res:
            ref = ref + base[:, choff:choff + cout].double()
        out = base.clone().cuda()
        ep = _lib.make_epilogue(bias=None if b is None else b.cuda(), noise=None if nz is None else nz.cuda(),
                                noise_strength=ns.cuda() if has_noise else None, noise_n=n, act=act, alpha=0.2, gain=gain,
                                residual=out if has_res else None)
        cv.conv_forward(x.cuda(), cv.pack_weights(wt.cuda()), stride=stride, pad=(pad, pad), in_scale=None if s_in is None else s_in.cuda(),
                        out_scale=None if s_out is None else s_out.cuda(), epilogue=ep, out=out, out_choff=choff)
        tag = (f"case {case}: x {tuple(x.shape)} w {tuple(wt.shape)} stride {stride} pad {pad} {act} bias {has_bias} noise {has_noise} "
               f"residual {has_res} in_scale {has_is} out_scale {has_os} slice {choff}/{ctotal}")
        assert rel_err(out[:, choff:choff + cout], ref) < 2e-4, tag
        keep = torch.ones(ctotal, dtype=torch.bool)
        keep[choff:choff + cout] = False
        assert torch.equal(out.cpu()[:, keep], base[:, keep]), tag      # the rest of the wider buffer is untouched

def test_winograd3_and_pointwise_random_shapes():
    """The two MFMA convolutions whose dispatch has the most branches, on shapes nobody picked: form-3 Winograd under each workgroup shape (11 / 21 / 12:
    the one-shot kernel's one-block and two-block forms with their different operand rings), K from one chunk to 80, ragged maps, every epilogue piece,
    channel slices; the 1x1 register-operand GEMM over its split-K / one-block-per-wave / 16-byte / ragged-row branches.  Against float64 torch convolutions."""
    from morphganformer_amd import _lib, conv as cv
    from oracle.ops_ref import bias_act_ref
    r = _rng(1212)
    f = lambda t: t.cuda().contiguous()
    try:
        for case in range(36):
            shape = (11, 21, 12)[case % 3]
            n = int(r.integers(1, 4))
            cin = 4 * int(r.integers(1, 41)) if case % 4 else int(r.choice([256, 288, 320]))
            cout = 32 * int(r.integers(1, 6)) * (2 if shape == 21 else 1)
            h, w = int(r.integers(3, 60)), int(r.integers(3, 100))
            torch.manual_seed(12000 + case + _OFFSET)
            x = torch.randn(n, cin, h, w)
            wt = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
            s, d = 1 + 0.3 * torch.randn(n, cin), 0.5 + torch.rand(n, cout)
            ref = torch.nn.functional.conv2d((x * s[:, :, None, None]).double(), wt.double(), padding=1) * d[:, :, None, None].double()
            kw, ep = {}, {}
            if r.integers(0, 2):
                bias = torch.randn(cout); ep["bias"] = f(bias)
            else:
                bias = None
            act = ["linear", "lrelu", "relu"][int(r.integers(0, 3))]
            alpha = float(r.uniform(0.05, 0.9)) if act == "lrelu" else None
            gain = float(r.uniform(0.5, 2.0))
            noise = torch.randn(n, h, w) if r.integers(0, 2) else None
            resid = torch.randn(n, cout, h, w) if r.integers(0, 2) else None
            want = ref.float()
            if noise is not None:
                strength = torch.tensor([float(r.uniform(0.1, 1.0))])
                want = want + noise[:, None] * strength
                ep.update(noise=f(noise), noise_strength=strength.cuda(), noise_n=n)
            want = bias_act_ref(want, bias, act=act, alpha=alpha, gain=gain)
            if resid is not None:
                want = want + resid
                ep["residual"] = f(resid)
            _lib.check(_lib.lib().mgf_winograd3_force_shape(shape))
            out = cv.winograd_forward(f(x), cv.winograd2_weights(f(wt), gain=1.0), in_scale=f(s), out_scale=f(d),
                                      epilogue=_lib.make_epilogue(act=act, alpha=0.0 if alpha is None else alpha, gain=gain, **ep))
            assert rel_err(out, want) < 3e-5, (case, shape, n, cin, cout, h, w, act, sorted(ep))
    finally:
        _lib.lib().mgf_winograd3_force_shape(0)
    for case in range(60):
        n = int(r.integers(1, 5))
        cin, cout = int(r.integers(1, 600)), int(r.integers(1, 300))
        if case % 3 == 0:
            cin, cout = int(r.choice([256, 384, 512])), int(r.choice([32, 64, 256, 512]))
        h, w = int(r.integers(1, 40)), int(r.integers(1, 70))
        if case % 7 == 0:
            h, w = int(r.integers(60, 130)), int(r.integers(60, 130))
        extra = int(r.integers(0, 3)) * 8
        choff = int(r.integers(0, extra + 1))
        torch.manual_seed(13000 + case + _OFFSET)
        x = torch.randn(n, cin, h, w)
        wt = torch.randn(cout, cin, 1, 1) / cin ** 0.5
        bias = torch.randn(cout)
        act = ["linear", "lrelu", "relu"][int(r.integers(0, 3))]
        want = bias_act_ref(torch.nn.functional.conv2d(x.double(), wt.double()).float(), bias, act=act, alpha=0.3 if act == "lrelu" else None, gain=1.0)
        out = torch.full((n, cout + extra, h, w), 5.0, device="cuda")
        kw = {}
        if r.integers(0, 2):
            resid = torch.randn(n, cout + extra, h, w)
            want = want + resid[:, choff:choff + cout]
            kw["residual"] = f(resid)
        cv.conv_forward(f(x), cv.pack_weights(f(wt)), out=out, out_choff=choff,
                        epilogue=_lib.make_epilogue(bias=f(bias), act=act, alpha=0.3 if act == "lrelu" else 0.0, gain=1.0, **kw))
        assert rel_err(out[:, choff:choff + cout], want) < 3e-5, (case, n, cin, cout, h, w, act, extra, choff)
        rest = torch.ones(cout + extra, dtype=torch.bool); rest[choff:choff + cout] = False
        assert bool((out[:, rest.cuda()] == 5.0).all()), (case, "wrote outside its channel slice")

def test_transposed_conv_random_shapes():
    """The stride-2 transposed 3x3 convolution of the up-sampling layers (persistent tap-list kernel with one chunk stream across tiles, its border launch,
    the single-launch form of small maps and few images) on random channel counts and map sizes at 1 .. 5 images -- the n >= 4 / n < 4 and 16 px / 128 px
    dispatch boundaries included -- against float64 torch.  (Its adjoint, the stride-2 convolution of the data gradient, has its cases in test_hip_ops.py.)"""
    import math
    from morphganformer_amd import conv as cv
    r = _rng(1414)
    for case in range(40):
        n = int(r.integers(1, 6))
        cin = 8 * int(r.integers(1, 33)) if case % 3 else int(r.integers(1, 70))
        cout = int(r.integers(1, 140))
        h, w = int(r.integers(2, 48)), int(r.integers(2, 72))
        if case % 8 == 0:
            h, w, cin, cout = int(r.integers(128, 150)), int(r.integers(128, 140)), 8 * int(r.integers(1, 5)), int(r.integers(1, 40))
        torch.manual_seed(14000 + case + _OFFSET)
        x = torch.randn(n, cin, h, w)
        wt = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
        s, d = 1 + 0.2 * torch.randn(n, cin), 1 + 0.2 * torch.randn(n, cout)
        ref = torch.nn.functional.conv_transpose2d((x * s[:, :, None, None]).double(), wt.transpose(0, 1).double(), stride=2) * d[:, :, None, None].double()
        out = cv.tconv3x3s2_forward(x.cuda(), cv.pack_weights(wt.cuda()), in_scale=s.cuda(), out_scale=d.cuda())
        assert tuple(out.shape) == tuple(ref.shape) and rel_err(out, ref) < 2e-5, (case, "tconv", n, cin, cout, h, w)

def test_projection_engine_random_option_combinations():
    """The literal loop's plumbing under option COMBINATIONS nobody wrote a test for (candidates per forward that do not divide the step count, graph replay
    and the two-stream pipeline, wing / adaptive wing with "no face" steps, LPIPS on or off with its coefficient, each pixel term, the v2 driver's averaged
    latent copies): every recorded loss against the same terms evaluated one candidate at a time through the public modules, the best
    step by the drivers' rule (strictly smaller than everything before, skipped steps never), the kept latent bit-exact."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, synthetic_landmarks
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.loss_ref import adaptive_wing_loss_ref, dssim_ref, psnr_ref, psnr_script_ref, wing_loss_ref
    cfg = TINY
    G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
    P = PerceptualLoss(net="squeeze", allow_random_backbone=True)
    r = _rng(1515)
    for case in range(14):
        steps = int(r.integers(3, 13))
        batch = int(r.choice([1, 2, 3, 4, 5, 7]))
        use_graph, pipeline = bool(r.integers(0, 2)), bool(r.integers(0, 2))
        copies = int(r.choice([1, 1, 3]))
        pixel = str(r.choice(["mse", "mse", "psnr", "dssim", "none"]))
        layout = str(r.choice(["script", "aligned"]))
        use_percept = bool(r.integers(0, 2)) or pixel == "none"
        wing = str(r.choice(["none", "wing", "awing"]))
        a = ProjectionArgs(step=steps, lamda=float(r.uniform(1e-3, 2e-2)) if wing != "awing" else 1e-5, beta=float(r.uniform(0.2, 2.0)),
                           percept_weight=float(r.uniform(0.3, 1.5)), min_loss_init=float(r.choice([1e9, 1e12])),      # (the tiny generator's images reach MSE 200: the drivers' 100 would end in the reference's IndexError)
                           pixel_term="mse" if pixel == "none" else pixel, psnr_layout=layout, latent_copies=copies)
        torch.manual_seed(15000 + case + _OFFSET)
        latent_mean = torch.randn(cfg.k, cfg.z_dim)
        latent_std = float(r.uniform(0.5, 2.0))
        eps = torch.randn(steps, 1, copies, cfg.k, cfg.z_dim) if copies > 1 else torch.randn(steps, 1, cfg.k, cfg.z_dim)
        target = G(torch.randn(1, cfg.k, cfg.z_dim).cuda(), None, noise_mode="const")[0].clamp(-1, 1).clone()
        kw = {}
        valid = np.ones(steps, np.int32)
        if wing != "none":
            lm_t, lm_s = synthetic_landmarks(steps, 64, 50 + case)
            if wing == "awing":
                lm_t, lm_s = lm_t / 64.0, lm_s / 64.0
            valid[r.integers(0, steps, int(r.integers(0, 3)))] = 0
            valid[int(r.integers(0, steps))] = 1
            kw.update(lm_target=lm_t, lm_steps=lm_s, lm_valid=valid, wing_kind=wing)
        tag = (case, steps, batch, use_graph, pipeline, copies, pixel, layout, use_percept, wing, valid.tolist())
        eng = ProjectionEngine(G, target, latent_mean.cuda(), latent_std, a, percept=P if use_percept else None, use_mse=pixel != "none",
                               eps=eps.cuda(), noise_mode="const", use_graph=use_graph, batch=batch, pipeline=pipeline, **kw)
        try:
            lat, bstep, bloss, losses = eng.run().result()
        except IndexError:
            raise AssertionError((tag, "no step improved", eng.losses.cpu().numpy(), a))
        want, zs = [], []
        tgt_np = target.cpu().numpy()
        for i in range(steps):
            sigma = np.float32(np.float32(latent_std) * np.float32(a.noise)) * np.float32(max(0, 1 - (i / steps) / a.noise_ramp) ** 2)
            if copies > 1:
                z = torch.mean(latent_mean[None, None].expand(1, copies, -1, -1) + eps[i] * float(sigma), 1)      # torch.mean over the copies (v2 driver :166)
            else:
                z = latent_mean[None] + eps[i] * float(sigma)
            zs.append(z)
            if not valid[i]:
                want.append(np.nan)
                continue
            img = G(z.cuda(), None, noise_mode="const")[0]
            v = 0.0
            if wing == "wing":
                v += a.lamda * float(wing_loss_ref(torch.from_numpy(lm_s[i]), torch.from_numpy(lm_t)))
            elif wing == "awing":
                v += a.lamda * float(adaptive_wing_loss_ref(torch.from_numpy(lm_s[i]), torch.from_numpy(lm_t)))
            if use_percept:
                v += a.percept_weight * float(P(img, target))
            if pixel == "mse":
                v += a.beta * float(torch.nn.functional.mse_loss(img, target))
            elif pixel == "psnr":
                v += a.beta * float((psnr_script_ref if layout == "script" else psnr_ref)(img.cpu().numpy(), tgt_np))
            elif pixel == "dssim":
                v += a.beta * float(dssim_ref(img.cpu().numpy()[0], tgt_np[0]))
            want.append(v)
        want = np.array(want)
        assert np.array_equal(np.isnan(losses), np.isnan(want)), tag
        ok = ~np.isnan(want)
        assert np.abs(losses[ok] - want[ok]).max() < 2e-4 * np.abs(want[ok]).max() + (1e-3 if pixel == "dssim" else 0.0), (tag, losses, want)
        best, cur = -1, a.min_loss_init
        for i in range(steps):
            if ok[i] and losses[i] < cur:
                best, cur = i, losses[i]
        assert bstep == best and bloss == float(losses[best]), (tag, bstep, best)
        assert torch.equal(lat.reshape(-1), zs[best].reshape(-1)), tag


@pytest.mark.parametrize("net", ["squeeze", "alex", "vgg"])
def test_lpips_random_non_square_sizes(net):
    """lpips.PerceptualLoss on image sizes nobody tuned a kernel for: non-square, odd, just above each backbone's minimum."""
    from morphganformer_amd.lpips import PerceptualLoss, random_backbone, random_squeeze_backbone
    from oracle.loss_ref import backbone_random, lpips_ref, squeeze_backbone_random
    r = _rng({"squeeze": 707, "alex": 708, "vgg": 709}[net])
    bb_np = random_squeeze_backbone(0) if net == "squeeze" else random_backbone(net, 0)
    bb = squeeze_backbone_random(0) if net == "squeeze" else backbone_random(net, 0)
    P = PerceptualLoss(model="net-lin", net=net, use_gpu=True, backbone_state=bb_np)
    lins = [l.cpu() for l in P.lins]
    lo = {"squeeze": 35, "alex": 70, "vgg": 33}[net]
    for case in range(6):
        h, w = int(r.integers(lo, 150)), int(r.integers(lo, 150))
        n = int(r.integers(1, 4))
        torch.manual_seed(6000 + case + _OFFSET)
        x0 = torch.rand(n, 3, h, w) * 2 - 1
        x1 = (x0[:1] + 0.3 * torch.randn(1, 3, h, w)).clamp(-1, 1)
        P.set_target(x1.cuda())
        out = torch.zeros(n, device="cuda")
        P.distance_into(out, x0.cuda())
        for i in range(n):
            ref = float(lpips_ref(bb, lins, x0[i:i + 1], x1, net=net))
            assert abs(float(out[i]) - ref) < 1e-3 * abs(ref), (net, h, w, n, i, float(out[i]), ref)


def test_warp_random_point_sets_are_byte_exact():
    """drivers.warp_morph on random landmark sets (random counts, jitters, image sides, non-square images) against the literal OpenCV
    transcription: integer / byte work, the bar is bit-exact."""
    from morphganformer_amd import drivers
    from oracle import warp_ref as W
    for case in range(8):
        rng = np.random.Generator(np.random.PCG64(900 + case))
        hh, ww = int(rng.integers(40, 120)), int(rng.integers(40, 120))
        n_inner = int(rng.integers(3, 30))
        img = rng.integers(0, 256, (hh, ww, 3)).astype(np.uint8)
        inner = np.stack([rng.integers(4, ww - 4, n_inner), rng.integers(4, hh - 4, n_inner)], axis=1)
        inner = np.unique(inner, axis=0)
        frame = np.array([[0, 0], [ww // 2, 0], [ww - 1, 0], [0, hh // 2], [ww - 1, hh // 2], [0, hh - 1], [ww // 2, hh - 1], [ww - 1, hh - 1]])
        jit = int(rng.integers(1, 6))
        p_src = np.concatenate([(inner + rng.integers(-jit, jit + 1, inner.shape)).clip(0, [ww - 1, hh - 1]), frame])
        p_dst = np.concatenate([inner, frame])
        label, recs, simp = drivers.warp_plan(p_src, p_dst, hh, ww)
        ref = W.warp_morph_ref(img.astype(np.float32), [tuple(p) for p in p_src], [tuple(p) for p in p_dst], simp)
        got = drivers.warp_morph_u8(img, p_src, p_dst)
        assert np.array_equal(got, np.uint8(ref)), (case, hh, ww, n_inner, jit)


@pytest.mark.parametrize("res,base,cmax,att,norm_g", [
    (32, 256, 16, 8, True), (64, 1024, 64, 6, False), (128, 2048, 48, 8, True), (256, 4096, 128, 7, True), (64, 512, 32, 3, True),
    (128, 16384, 256, 8, False), (16, 512, 64, 8, True)])
def test_generator_random_configurations_vs_oracle(res, base, cmax, att, norm_g):
    """Generator configurations besides the two the goldens pin (run_network.py:61-77,243-283: resolution, channel_base / channel_max,
    the attention range, normalize_global): image, latents and every noise mode's plumbing against the CPU oracle."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import GeneratorConfig, make_state_dict
    from oracle.generator_ref import generator_ref, to_torch_state
    cfg = GeneratorConfig(img_resolution=res, channel_base=base, channel_max=cmax, attn_max_log2res=att, normalize_global=norm_g)
    sd = make_state_dict(cfg, seed=res + cmax + _OFFSET % 9973)          # (soak: other weights too)
    G = Generator(sd, cfg, "cuda", max_batch=3)
    torch.manual_seed(res * 7 + cmax + _OFFSET)
    z = torch.randn(3, cfg.k, cfg.z_dim)
    taps = {}
    ref = generator_ref(to_torch_state(sd), z, cfg, "const", taps=taps)
    img, att = G(z.cuda(), None, noise_mode="const", return_att=True, att_format="maps")
    assert tuple(img.shape) == (3, 3, res, res)
    assert rel_err(img, ref) < 1e-3
    # the integer gate on every attention layer of the configuration: the latent a pixel is assigned to, exact wherever the oracle's top two are 1e-4 apart
    layers = [k[:-len(":probs")] for k in taps if k.endswith(":probs")]
    assert sorted(layers) == sorted(att) and len(layers) >= 1
    for key in layers:
        probs, argmax = att[key]
        want = taps[key + ":probs"].detach().numpy().reshape(probs.shape)
        assert np.abs(probs.cpu().numpy() - want).max() < 1e-4, key
        top2 = np.sort(want, axis=-1)[..., -2:]
        decided = (top2[..., 1] - top2[..., 0]) > 1e-4
        assert decided.mean() > 0.97 and np.array_equal(argmax.cpu().numpy()[decided], want.argmax(-1)[decided]), key
    ref0 = generator_ref(to_torch_state(sd), z[:2], cfg, "none", truncation_psi=0.6)
    img0 = G(z[:2].cuda(), None, noise_mode="none", truncation_psi=0.6)[0]
    assert rel_err(img0, ref0) < 1e-3


@pytest.mark.parametrize("res,base,cmax,att,norm_g", [(32, 256, 16, 8, True), (128, 2048, 48, 6, False), (256, 4096, 96, 8, True),
                                                       (64, 16384, 256, 4, True)])
def test_generator_gradient_random_configurations_vs_autograd(res, base, cmax, att, norm_g):
    """d(loss)/dz through other generator configurations than the goldens' (gradient mode) against torch autograd
    through the CPU oracle; gate as in test_hip_gradient.py: 1e-3 of max |gradient|."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.grad import GeneratorGrad
    from morphganformer_amd.synth_weights import GeneratorConfig, make_state_dict
    from oracle.generator_ref import generator_ref, to_torch_state
    cfg = GeneratorConfig(img_resolution=res, channel_base=base, channel_max=cmax, attn_max_log2res=att, normalize_global=norm_g)
    sd = make_state_dict(cfg, seed=res + cmax + 1 + _OFFSET % 9973)
    gg = GeneratorGrad(Generator(sd, cfg, "cuda", max_batch=2))
    torch.manual_seed(res + 3 * cmax + _OFFSET)
    z = torch.randn(2, cfg.k, cfg.z_dim, requires_grad=True)
    target = torch.randn(2, 3, res, res) * 0.5
    img_ref = generator_ref(to_torch_state(sd), z, cfg, "const")
    (dz_ref,) = torch.autograd.grad((img_ref - target).square().mean(dim=(1, 2, 3)).sum(), z)
    img = gg.forward(z.detach().cuda(), noise_mode="const")
    assert rel_err(img, img_ref) < 1e-3
    dz = gg.backward(2.0 * (img - target.cuda()) / target[0].numel())
    assert rel_err(dz, dz_ref) < 1e-3


def test_second_order_gradients_random():
    """Double backward of the two plugin operators (bias_act.py:137-198: the grad=2 kernel; upfirdn2d.py:237-256: self-application) on random
    shapes, against torch's own double backward through the oracle's float64 forward."""
    from morphganformer_amd.torch_utils.ops import bias_act, upfirdn2d
    from oracle.ops_ref import bias_act_ref, upfirdn2d_ref
    r = _rng(808)
    smooth = ["tanh", "sigmoid", "softplus", "swish", "elu", "selu", "linear", "lrelu"]
    for case in range(60):
        rank = int(r.integers(2, 5))
        shape = [int(v) for v in r.integers(1, 8, rank)]
        dim = int(r.integers(0, rank))
        act = smooth[int(r.integers(0, len(smooth)))]
        gain = float(r.uniform(0.5, 2.0))
        torch.manual_seed(7000 + case + _OFFSET)
        x, b = torch.randn(*shape), torch.randn(shape[dim])
        gy, ggx = torch.randn(*shape), torch.randn(*shape)
        xr = x.double().requires_grad_(True)
        yr = bias_act_ref(xr, b.double(), dim=dim, act=act, gain=gain)
        (dxr,) = torch.autograd.grad(yr, xr, gy.double(), create_graph=True)
        xg = x.cuda().requires_grad_(True)
        yg = bias_act.bias_act(xg, b.cuda(), dim=dim, act=act, gain=gain)
        (dxg,) = torch.autograd.grad(yg, xg, gy.cuda(), create_graph=True)
        tag = f"bias_act case {case}: {shape} dim {dim} {act}"
        assert rel_err(dxg, dxr) < 1e-5, tag
        if dxr.requires_grad and dxr.grad_fn is not None and act not in ("linear", "lrelu"):
            (d2r,) = torch.autograd.grad(dxr, xr, ggx.double())
            (d2g,) = torch.autograd.grad(dxg, xg, ggx.cuda())
            assert rel_err(d2g, d2r) < 2e-5, tag
    for case in range(40):
        n, c = int(r.integers(1, 3)), int(r.integers(1, 4))
        h, w = int(r.integers(3, 30)), int(r.integers(3, 40))
        up, down = int(r.integers(1, 3)), int(r.integers(1, 3))
        ft = int(r.integers(1, 5))
        pad = [int(v) for v in r.integers(0, 3, 4)]
        if h * up + pad[2] + pad[3] < ft or w * up + pad[0] + pad[1] < ft:
            continue
        torch.manual_seed(8000 + case + _OFFSET)
        x, f = torch.randn(n, c, h, w), torch.rand(ft, ft) + 0.1
        xr = x.double().requires_grad_(True)
        yr = upfirdn2d_ref(xr, f.double(), up=up, down=down, padding=pad, gain=1.5)
        gy = torch.randn(yr.shape)
        (dxr,) = torch.autograd.grad(yr, xr, gy.double(), create_graph=True)
        # the operator is linear in x: its gradient does not depend on x, the double backward w.r.t. gy is the forward operator again
        gyg = gy.cuda().requires_grad_(True)
        xg = x.cuda().requires_grad_(True)
        yg = upfirdn2d.upfirdn2d(xg, f.cuda(), up=up, down=down, padding=pad, gain=1.5)
        (dxg,) = torch.autograd.grad(yg, xg, gyg, create_graph=True)
        v = torch.randn(n, c, h, w)
        (back,) = torch.autograd.grad(dxg, gyg, v.cuda())
        want = upfirdn2d_ref(v.double(), f.double(), up=up, down=down, padding=pad, gain=1.5)
        tag = f"upfirdn2d case {case}: x {tuple(x.shape)} up {up} down {down} f {ft} pad {pad}"
        assert rel_err(dxg, dxr) < 1e-5 and rel_err(back, want) < 1e-5, tag


@pytest.mark.parametrize("net", ["squeeze", "alex", "vgg"])
def test_lpips_gradient_random_non_square_sizes(net):
    """d LPIPS / d image (gradient mode) on non-square, odd image sizes against autograd through the oracle.
    (The gate holds AWAY from ReLU / max-pool ties: an input with one unit within rounding of its decision boundary gets that unit's whole
    contribution or none -- seen once while these cases were drawn, vgg 109 x 66: 2.7e-3 of max |gradient| over one receptive field against
    float64, 2.7e-6 after perturbing the input by 1e-5.  The seeds below have no such unit.)"""
    import os
    from morphganformer_amd.lpips import PerceptualLoss, WEIGHTS_DIR
    from oracle.loss_ref import backbone_random, lpips_ref, squeeze_backbone_random
    r = _rng({"squeeze": 811, "alex": 812, "vgg": 814}[net])
    bb = squeeze_backbone_random(0) if net == "squeeze" else backbone_random(net, 0)
    lin = np.load(os.path.join(WEIGHTS_DIR, f"lpips_lin_{net}.npz"))
    lins = [torch.from_numpy(lin[f"lin{i}"]).float().reshape(-1) for i in range(len(lin.files))]
    pl = PerceptualLoss(net=net, allow_random_backbone=True)
    lo = {"squeeze": 35, "alex": 70, "vgg": 33}[net]
    for case in range(4):
        h, w = int(r.integers(lo, 120)), int(r.integers(lo, 120))
        n = int(r.integers(1, 3))
        torch.manual_seed(9000 + case + _OFFSET)
        pred = (torch.rand(n, 3, h, w) * 2 - 1).requires_grad_(True)
        target = torch.rand(1, 3, h, w) * 2 - 1
        val = lpips_ref(bb, lins, pred, target.expand(n, -1, -1, -1), net=net)
        (ref,) = torch.autograd.grad(val.sum(), pred)
        pl.set_target(target.cuda())
        out = torch.empty(n, device="cuda")
        pl.distance_into(out, pred.detach().cuda(), keep_taps=True)
        assert rel_err(out, val.reshape(n)) < 1e-3, (net, h, w, n)
        dimg = torch.zeros(n, 3, h, w, device="cuda")
        pl.grad_into(dimg, scale=1.0)
        assert rel_err(dimg, ref) < 1e-3, (net, h, w, n)


def test_facenet_random_non_square_sizes():
    """The InceptionResnetV1 embedder on random image sizes from just above its minimum (75) upwards, non-square and odd: every stride-2 /
    valid-padding stage hits different ragged edges.  Embedding <= 1e-3 against the oracle (unit norm), seeded weights."""
    from morphganformer_amd.facenet import InceptionResnetV1Embedder, random_state
    from oracle.embed_ref import inception_resnet_v1_ref
    r = _rng(909)
    sd_np = random_state(5)
    sd_t = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    net = InceptionResnetV1Embedder(sd_np, n=1)
    for case in range(5):
        h, w = int(r.integers(75, 240)), int(r.integers(75, 240))
        n = int(r.integers(1, 3))
        torch.manual_seed(9500 + case + _OFFSET)
        x = torch.rand(n, 3, h, w) * 2 - 1
        emb = net(x.cuda()).cpu()
        with torch.no_grad():
            want = inception_resnet_v1_ref(sd_t, x, {})
        assert tuple(emb.shape) == (n, 512), (h, w, n)
        assert float((emb - want).abs().max()) < 1e-3 * float(want.abs().max()), (h, w, n)
    with pytest.raises(Exception):
        net(torch.zeros(1, 3, 60, 60, device="cuda"))            # below the network's minimum: refused, not garbage
