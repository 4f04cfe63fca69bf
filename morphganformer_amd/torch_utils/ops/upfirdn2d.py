"""upfirdn2d operator family -- same Python contract as the reference's torch_utils/ops/upfirdn2d.py
(setup_filter :64, upfirdn2d :112, filter2d :264, upsample2d :300, downsample2d :339), executed by the gfx950 kernels behind
`mgf_upfirdn2d` (include/mgf.h).  Gradients of any order come from self-application with up/down swapped and the filter
flipped, as in the reference (:237-256)."""
from __future__ import annotations

import numpy as np
import torch

from ... import _lib


def _as_ints(value, counts, what):
    """An int or a sequence whose length is one of `counts`, as a list of Python ints (error behaviour of the reference: assert)."""
    seq = [value] if isinstance(value, int) else list(value)
    assert len(seq) in counts and all(isinstance(v, int) for v in seq), f"bad {what}: {value!r}"
    return seq


def _parse_scaling(scaling):
    """up / down factor: k or (kx, ky), each >= 1."""
    seq = _as_ints(scaling, (1, 2), "scaling")
    sx, sy = seq * 2 if len(seq) == 1 else seq
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    """padding: p | (px, py) | (px0, px1, py0, py1) -> (px0, px1, py0, py1); negative values crop."""
    seq = _as_ints(padding, (1, 2, 4), "padding")
    if len(seq) == 1:
        seq = seq * 4
    elif len(seq) == 2:
        seq = [seq[0], seq[0], seq[1], seq[1]]
    return tuple(seq)


def _get_filter_size(f):
    """(width, height) of a prepared filter; None stands for the 1x1 identity."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device("cpu"), normalize=True, flip_filter=False, gain=1, separable=None):
    """Prepare a float32 FIR filter: 1-D taps of fewer than 8 elements become their 2-D outer product."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in (0, 1, 2) and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def _launch(x, f2d, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain, epilogue=None):
    n, c, h, w = x.shape
    fh, fw = f2d.shape
    oh = (h * upy + py0 + py1 - fh + downy) // downy
    ow = (w * upx + px0 + px1 - fw + downx) // downx
    if oh < 1 or ow < 1:
        raise _lib.MgfError("upfirdn2d: output would be empty")
    cl = x.ndim == 4 and x.stride(1) == 1 and c > 1 and not x.is_contiguous()
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if cl else torch.contiguous_format)
    sx, sy = x.stride(), y.stride()
    rc = _lib.lib().mgf_upfirdn2d(y.data_ptr(), x.data_ptr(), f2d.data_ptr(), _lib.dtype_id(x.dtype), n, c, h, w,
                                  sx[0], sx[1], sx[2], sx[3], oh, ow, sy[0], sy[1], sy[2], sy[3], fh, fw, upx, upy, downx,
                                  downy, px0, px1, py0, py1, int(flip), float(gain), epilogue, _lib.stream_ptr())
    _lib.check(rc, "upfirdn2d")
    return y


def _run(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain):
    """Rank-1 (separable) filters run as two passes, like the plugin wrapper (upfirdn2d.py:223-229)."""
    if not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
        x = x.contiguous()
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    if f.dtype != torch.float32 or f.device != x.device:
        raise _lib.MgfError("upfirdn2d: f must be a float32 tensor on x's device")
    f = f.contiguous()
    if f.ndim == 2:
        return _launch(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain)
    y = _launch(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip, np.sqrt(gain))
    return _launch(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip, np.sqrt(gain))


_cache = {}


def _make_op(up, down, padding, flip_filter, gain):
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    key = (upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
    if key in _cache:
        return _cache[key]

    class Upfirdn2dHip(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, f):
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            y = _run(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
            ctx.save_for_backward(f)
            ctx.x_shape = x.shape
            return y

        @staticmethod
        def backward(ctx, dy):
            (f,) = ctx.saved_tensors
            _, _, ih, iw = ctx.x_shape
            _, _, oh, ow = dy.shape
            fw, fh = _get_filter_size(f)
            p = [fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1]
            dx = None
            if ctx.needs_input_grad[0]:
                dx = _make_op([downx, downy], [upx, upy], p, not flip_filter, gain).apply(dy, f)
            assert not ctx.needs_input_grad[1]
            return dx, None

    _cache[key] = Upfirdn2dHip
    return Upfirdn2dHip


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Pad, upsample, filter and downsample a batch of 2-D images (reference docstring: upfirdn2d.py:112-150)."""
    assert isinstance(x, torch.Tensor)
    if impl == "ref":
        raise NotImplementedError("impl='ref' is not part of the MI355X package; use oracle.ops_ref.upfirdn2d_ref in tests")
    assert impl in ("cuda", "hip")
    _lib.require_gpu(x, f)
    return _make_op(up, down, padding, flip_filter, gain).apply(x, f)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl="cuda"):
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
