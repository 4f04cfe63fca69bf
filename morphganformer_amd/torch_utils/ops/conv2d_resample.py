"""conv2d_resample operator -- same Python contract as the reference's torch_utils/ops/conv2d_resample.py:51-146, executed
by the FP32-MFMA tap-list convolution (`mgf_conv_taps_f32`) and `mgf_upfirdn2d`.

Differentiable like the reference's (whose convolutions go through torch_utils/ops/conv2d_gradfix.py:50-162): every convolution is one
`torch.autograd.Function` whose backward is THE SAME function with the `adjoint` flag flipped -- a stride-1 correlation's gradient is the
correlation with the flipped, channel-transposed kernel, a strided correlation's gradient is the transposed convolution (the 3x3 / stride-2
one of the up-sampling layers runs on the kernel the forward of `up=2` uses, and vice versa) -- so gradients of any order with respect to `x`
compose with `upfirdn2d`'s self-application (upfirdn2d.py:237-256).  The modulation scales of `modulated_conv2d` receive first-order
gradients from the gradient-mode kernels (`mgf_style_grad_f32`, `mgf_channel_dot_f32`, `mgf_demod_bwd_f32`).  Weights are constants on
this path (the projection loop optimises the latent, never the generator): a weight that requires a gradient is
REFUSED, as are groups > 1 and dtypes other than float32 -- nothing here ever returns a silently detached tensor.
Weights are re-packed on every call here; the synthesis engine packs them once per checkpoint.
"""
from __future__ import annotations

import torch

from ... import _lib
from ... import conv as _conv
from . import upfirdn2d as _up


class _ConvOp:
    """One linear map y = out_scale * corr(w, in_scale * x) (stride, symmetric zero padding; `w` [co, ci, kh, kw] in correlation
    orientation) and its adjoint, on the library's kernels.  Packed forms are built on first use."""

    def __init__(self, w_corr, stride, pad):
        self.w = w_corr.detach().contiguous().float()
        self.co, self.ci, self.kh, self.kw = self.w.shape
        self.stride, self.pad = int(stride), (int(pad[0]), int(pad[1]))
        self._pc = None
        self._pcT = {}

    @property
    def pc(self):
        if self._pc is None:
            self._pc = _conv.pack_weights(self.w)
        return self._pc

    def pcT(self, flip):
        if flip not in self._pcT:
            self._pcT[flip] = _conv.transpose_packed(self.pc, flip=flip)
        return self._pcT[flip]

    @classmethod
    def from_transposed_pack(cls, w, flip_weight):
        """The stride-2 3x3 correlation whose ADJOINT is conv_transpose2d(x, w.transpose(0, 1)) as conv2d_resample.py:117-123 calls it:
        the reference hands conv_transpose2d un-flipped weights when flip_weight is False, flipped ones when it is True."""
        wf = w.flip([2, 3]) if flip_weight else w
        return cls(wf.transpose(0, 1), 2, (0, 0))              # [ci_of_x -> co] as the adjoint sees it: co and ci trade places

    def fwd(self, x, s_in, s_out):
        if self.kh * self.kw > _lib.MAX_TAPS:
            # more taps than one launch carries (the generator has none; the contract takes any kernel): chained <= 9-tap launches
            if s_in is not None or s_out is not None:
                raise _lib.MgfError("conv2d_resample: modulation with a kernel of more than 9 taps is not supported by the HIP path")
            return _conv.conv_large_forward(x, self.w, None, self.stride, self.pad, act="linear")
        return _conv.conv_forward(x, self.pc, stride=self.stride, pad=self.pad, in_scale=s_in, out_scale=s_out)

    def adj(self, g, s_in, s_out, in_hw):
        """x-shaped result [n, ci, *in_hw] = s_out * corr^T(w, s_in * g)."""
        H, W = in_hw
        py, px = self.pad
        n, _, hy, wy = g.shape
        if self.stride == 1:
            qy, qx = self.kh - 1 - py, self.kw - 1 - px
            if qy < 0 or qx < 0:                             # padding beyond the kernel: those output rows / columns never saw the input
                cy, cx = max(-qy, 0), max(-qx, 0)
                g = g[:, :, cy:hy - cy, cx:wy - cx].contiguous()
                qy, qx = max(qy, 0), max(qx, 0)
            if self.kh * self.kw > _lib.MAX_TAPS:
                if s_in is not None or s_out is not None:
                    raise _lib.MgfError("conv2d_resample: modulation with a kernel of more than 9 taps is not supported by the HIP path")
                return _conv.conv_large_forward(g, self.w.permute(1, 0, 2, 3).flip(2, 3).contiguous(), None, 1, (qy, qx), act="linear")
            return _conv.conv_forward(g, self.pcT(True), pad=(qy, qx), in_scale=s_in, out_scale=s_out)
        if self.stride == 2 and (self.kh, self.kw) == (3, 3) and self.pad == (0, 0) and (H, W) == (2 * hy + 1, 2 * wy + 1):
            # the up-sampling layers' transposed convolution at its own FLOP count (a view of a padded-pitch workspace)
            return _conv.tconv3x3s2_forward(g, self.pcT(False), in_scale=s_in, out_scale=s_out)
        return _conv.conv_strided_dgrad(g, self.w, self.stride, self.pad, (H, W), in_scale=s_in, out_scale=s_out)


def _partial_sums(launch, n, c, hw, device):
    chunks = int(_lib.lib().mgf_bwd_chunks(hw))
    part = torch.empty([n, c, chunks], dtype=torch.float32, device=device)
    launch(part)
    return part.sum(dim=-1)


class _LinConv(torch.autograd.Function):
    """y = op(x) or op^T(x) with the modulation scales in the launch.  d/dx: the same Function with `adjoint` flipped (any order);
    d/d(scales): first order, from the gradient-mode reduction kernels."""

    @staticmethod
    def forward(ctx, x, s_in, s_out, op, adjoint, in_hw):
        x = x.contiguous()
        y = op.adj(x, s_in, s_out, in_hw) if adjoint else op.fwd(x, s_in, s_out)
        ctx.op, ctx.adjoint, ctx.x_hw = op, adjoint, tuple(x.shape[2:])
        need_si = s_in is not None and s_in.requires_grad
        need_so = s_out is not None and s_out.requires_grad
        ctx.save_for_backward(x if need_si else None, s_in, s_out, y if need_so else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, s_in, s_out, y = ctx.saved_tensors
        op, adjoint = ctx.op, ctx.adjoint
        need_x, need_si, need_so = ctx.needs_input_grad[:3]
        gx = gsi = gso = None
        if (need_si or need_so) and torch.is_grad_enabled():
            raise _lib.MgfError("modulated_conv2d: second-order gradients with respect to the styles are not implemented on the HIP path "
                                "(first order only; gradients with respect to x compose to any order)")
        gy = gy.contiguous()
        n = gy.shape[0]
        L, st = _lib.lib(), _lib.stream_ptr()
        if need_si:
            # g = the data gradient BEFORE the style scale; then dx = s g and ds = <x, g> per (sample, channel) in one pass
            g = (op.fwd(gy, s_out, None) if adjoint else op.adj(gy, s_out, None, ctx.x_hw)).contiguous()
            c, hw = x.shape[1], x.shape[2] * x.shape[3]
            gx = torch.empty_like(x)
            gsi = _partial_sums(lambda part: _lib.check(L.mgf_style_grad_f32(part.data_ptr(), gx.data_ptr(), x.data_ptr(), g.data_ptr(),
                                                                             s_in.data_ptr(), n, c, hw, 0, st), "style_grad"),
                                n, c, hw, x.device)
            if not need_x:
                gx = None
        elif need_x:
            gx = _LinConv.apply(gy, s_out, s_in, op, not adjoint, ctx.x_hw)
        if need_so:
            yc = y.contiguous()
            c, hw = yc.shape[1], yc.shape[2] * yc.shape[3]
            dot = _partial_sums(lambda part: _lib.check(L.mgf_channel_dot_f32(part.data_ptr(), gy.data_ptr(), yc.data_ptr(), n, c, hw, st),
                                                        "channel_dot"), n, c, hw, yc.device)
            gso = dot / s_out                                # y = s_out * (...): <gy, y> / s_out (a demodulation coefficient is never 0)
        return gx, gsi, gso, None, None, None


class _Demod(torch.autograd.Function):
    """d[n, co] = rsqrt(sum_ci wsq[co, ci] s[n, ci]^2 + 1e-8)  (networks.py:288-291 with the weights constant)."""

    @staticmethod
    def forward(ctx, s, wsq):
        n, cin = s.shape
        d = torch.empty([n, wsq.shape[0]], dtype=torch.float32, device=s.device)
        _lib.check(_lib.lib().mgf_demod_f32(d.data_ptr(), s.data_ptr(), wsq.data_ptr(), n, cin, wsq.shape[0], _lib.stream_ptr()), "demod")
        ctx.save_for_backward(s, wsq, d)
        return d

    @staticmethod
    def backward(ctx, dd):
        s, wsq, d = ctx.saved_tensors
        if torch.is_grad_enabled():
            raise _lib.MgfError("modulated_conv2d: second-order gradients through the demodulation are not implemented on the HIP path")
        ds = torch.empty_like(s)
        _lib.check(_lib.lib().mgf_demod_bwd_f32(ds.data_ptr(), dd.contiguous().data_ptr(), d.data_ptr(), s.data_ptr(), wsq.data_ptr(),
                                                s.shape[0], s.shape[1], wsq.shape[0], _lib.stream_ptr()), "demod_bwd")
        return ds, None


def _conv2d(x, w, stride=1, padding=(0, 0), flip_weight=True, in_scale=None, out_scale=None):
    """flip_weight=True is correlation (torch.nn.functional.conv2d), False is true convolution (conv2d_resample.py:27-28)."""
    op = _ConvOp(w if flip_weight else w.flip([2, 3]), stride, padding)
    return _LinConv.apply(x, in_scale, out_scale, op, False, None)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False,
                    in_scale=None, out_scale=None):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    _lib.require_gpu(x, w, f)
    if groups != 1:
        raise _lib.MgfError("conv2d_resample: groups > 1 is not supported by the HIP path (use modulated_conv2d)")
    if x.dtype != torch.float32:
        raise _lib.MgfError("conv2d_resample: the MFMA path is float32 only")
    if w.requires_grad and torch.is_grad_enabled():
        raise _lib.MgfError("conv2d_resample: the weights require a gradient, but the HIP path treats them as constants (it serves the latent "
                            "projection, which never trains the generator): call G.requires_grad_(False) / pass w.detach()")
    co, ci, kh, kw = w.shape
    fw, fh = _up._get_filter_size(f)
    px0, px1, py0, py1 = _up._parse_padding(padding)
    if up > 1:
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2

    if kw == 1 and kh == 1 and down > 1 and up == 1:
        x = _up.upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d(x, w, flip_weight=flip_weight, in_scale=in_scale, out_scale=out_scale)
    if kw == 1 and kh == 1 and up > 1 and down == 1:
        x = _conv2d(x, w, flip_weight=flip_weight, in_scale=in_scale, out_scale=out_scale)
        return _up.upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:
        x = _up.upfirdn2d(x, f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d(x, w, stride=down, flip_weight=flip_weight, in_scale=in_scale, out_scale=out_scale)
    if up == 2 and kh == 3 and kw == 3:
        # stride-2 transposed conv at its own FLOP count, then the FIR (conv2d_resample.py:117-134): the adjoint of a stride-2 correlation
        op = _ConvOp.from_transposed_pack(w, flip_weight)
        t = _LinConv.apply(x, in_scale, out_scale, op, True, (2 * x.shape[2] + 1, 2 * x.shape[3] + 1))
        px0 -= kw - 1; px1 -= kw - up; py0 -= kh - 1; py1 -= kh - up
        x = _up.upfirdn2d(t, f, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = _up.upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x
    if up == 1 and down == 1 and px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:
        return _conv2d(x, w, padding=(py0, px0), flip_weight=flip_weight, in_scale=in_scale, out_scale=out_scale)
    # generic ordering: upsample+pad -> conv -> downsample
    x = _up.upfirdn2d(x, f if up > 1 else None, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    x = _conv2d(x, w, flip_weight=flip_weight, in_scale=in_scale, out_scale=out_scale)
    if down > 1:
        x = _up.upfirdn2d(x, f, down=down, flip_filter=flip_filter)
    return x


def _add_noise(y, noise):
    """y + noise (networks.py:323-324), noise broadcast over the channels.  Without a graph to keep: one pass of the library's own
    upfirdn2d (identity filter) with the noise port of its epilogue; with one -- or a noise shape the port does not take -- torch's add,
    which autograd understands."""
    n, c, h, w = y.shape
    per_sample = noise.numel() == n * h * w and tuple(noise.shape[-2:]) == (h, w)
    shared = noise.numel() == h * w and tuple(noise.shape[-2:]) == (h, w)
    if (y.requires_grad or noise.requires_grad) and torch.is_grad_enabled() or not (per_sample or shared) or noise.dtype != torch.float32:
        return y + noise
    nz = noise.contiguous()
    one = torch.ones(1, dtype=torch.float32, device=y.device)
    ident = torch.ones([1, 1], dtype=torch.float32, device=y.device)
    ep = _lib.make_epilogue(noise=nz, noise_strength=one, noise_n=n if per_sample and n > 1 else 1)
    return _conv.upfirdn_into(torch.empty_like(y), y.contiguous(), ident, epilogue=ep)


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_kernel=None, demodulate=True,
                     flip_weight=True, fused_modconv=True, modulate=True):
    """training/networks.py:253-328.  The per-sample weights w*s*d are never materialised: s scales the input channels as
    they are staged into LDS and d scales the accumulators (exact in real arithmetic, re-associated in float32).  Differentiable with
    respect to x (any order) and the styles (first order, through both the modulation and the demodulation)."""
    _lib.require_gpu(x, weight, styles, noise)
    if not modulate:
        y = conv2d_resample(x, weight, f=resample_kernel, up=up, padding=padding, flip_weight=flip_weight)
        return _add_noise(y, noise) if noise is not None else y
    n = x.shape[0]
    co, ci, kh, kw = weight.shape
    assert styles.shape == (n, ci)
    if weight.requires_grad and torch.is_grad_enabled():
        raise _lib.MgfError("modulated_conv2d: the weights require a gradient, but the HIP path treats them as constants: "
                            "call G.requires_grad_(False) / pass weight.detach()")
    s = styles.contiguous().float()
    d = None
    if demodulate:
        wsq = _conv.pack_weights(weight.detach(), want_wsq=True).wsq                  # [co, ci] = sum over the taps of w^2
        d = _Demod.apply(s, wsq)                                                       # [n, co]
    y = conv2d_resample(x, weight, f=resample_kernel, up=up, down=down, padding=padding, flip_weight=flip_weight,
                        in_scale=s, out_scale=d)
    if noise is not None:
        y = _add_noise(y, noise)
    return y
