"""conv2d_resample operator -- same Python contract as the reference's torch_utils/ops/conv2d_resample.py:51-146, executed
by the FP32-MFMA tap-list convolution (`mgf_conv_taps_f32`) and `mgf_upfirdn2d`.

Inference-only (the projection loop of the reference never back-propagates through it); groups == 1
(the per-sample `groups=batch` trick of modulated_conv2d is replaced by in-register modulation, see modulated_conv2d below).
Weights are re-packed on every call here; the synthesis engine packs them once per checkpoint.
"""
from __future__ import annotations

import torch

from ... import _lib
from ... import conv as _conv
from . import upfirdn2d as _up


def _conv2d(x, w, stride=1, padding=(0, 0), flip_weight=True, in_scale=None, out_scale=None):
    """flip_weight=True is correlation (torch.nn.functional.conv2d), False is true convolution (conv2d_resample.py:27-28)."""
    if w.shape[2] * w.shape[3] > _lib.MAX_TAPS:
        # more taps than one launch carries (the generator has none; the contract takes any kernel): chained <= 9-tap launches
        if in_scale is not None or out_scale is not None:
            raise _lib.MgfError("conv2d_resample: modulation with a kernel of more than 9 taps is not supported by the HIP path")
        return _conv.conv_large_forward(x.contiguous(), w if flip_weight else w.flip([2, 3]), None, stride, tuple(padding), act="linear")
    pc = _conv.pack_weights(w, flip=not flip_weight)
    return _conv.conv_forward(x.contiguous(), pc, stride=stride, pad=padding, in_scale=in_scale, out_scale=out_scale)


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
        # stride-2 transposed conv at its own FLOP count, then the FIR (conv2d_resample.py:117-134).  The reference's
        # conv_transpose2d sees un-flipped weights when flip_weight is False, flipped ones when it is True.
        pc = _conv.pack_weights(w, flip=flip_weight)
        t = _conv.tconv3x3s2_forward(x.contiguous(), pc, in_scale=in_scale, out_scale=out_scale)
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


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_kernel=None, demodulate=True,
                     flip_weight=True, fused_modconv=True, modulate=True):
    """training/networks.py:253-328.  The per-sample weights w*s*d are never materialised: s scales the input channels as
    they are staged into LDS and d scales the accumulators (exact in real arithmetic, re-associated in float32)."""
    _lib.require_gpu(x, weight, styles, noise)
    if not modulate:
        y = conv2d_resample(x, weight, f=resample_kernel, up=up, padding=padding, flip_weight=flip_weight)
        return y.add_(noise) if noise is not None else y
    n = x.shape[0]
    co, ci, kh, kw = weight.shape
    assert styles.shape == (n, ci)
    s = styles.contiguous().float()
    d = None
    if demodulate:
        wsq = weight.float().square().sum(dim=[2, 3])                       # [co, ci]
        d = torch.rsqrt(s.square() @ wsq.t() + 1e-8).contiguous()           # [n, co]
    y = conv2d_resample(x, weight, f=resample_kernel, up=up, down=down, padding=padding, flip_weight=flip_weight,
                        in_scale=s, out_scale=d)
    if noise is not None:
        y = y.add_(noise)
    return y
