"""bias_act operator -- same Python contract as the reference's torch_utils/ops/bias_act.py:47-81, executed by the
gfx950 kernel behind `mgf_bias_act` (include/mgf.h).  First and second order gradients are provided the same way the
reference does (bias_act.py:137-198): the kernel's grad=1 / grad=2 forms.

impl='cuda' (reference spelling) and impl='hip' both mean the HIP kernel.  impl='ref' is deliberately unavailable in the
product package: the CPU restatement lives in oracle/ops_ref.py and is test infrastructure.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch

from ... import _lib

activation_funcs = {
    "linear":   SimpleNamespace(def_alpha=0,   def_gain=1,            cuda_idx=1, ref="",  has_2nd_grad=False),
    "relu":     SimpleNamespace(def_alpha=0,   def_gain=math.sqrt(2), cuda_idx=2, ref="y", has_2nd_grad=False),
    "lrelu":    SimpleNamespace(def_alpha=0.2, def_gain=math.sqrt(2), cuda_idx=3, ref="y", has_2nd_grad=False),
    "tanh":     SimpleNamespace(def_alpha=0,   def_gain=1,            cuda_idx=4, ref="y", has_2nd_grad=True),
    "sigmoid":  SimpleNamespace(def_alpha=0,   def_gain=1,            cuda_idx=5, ref="y", has_2nd_grad=True),
    "elu":      SimpleNamespace(def_alpha=0,   def_gain=1,            cuda_idx=6, ref="y", has_2nd_grad=True),
    "selu":     SimpleNamespace(def_alpha=0,   def_gain=1,            cuda_idx=7, ref="y", has_2nd_grad=True),
    "softplus": SimpleNamespace(def_alpha=0,   def_gain=1,            cuda_idx=8, ref="y", has_2nd_grad=True),
    "swish":    SimpleNamespace(def_alpha=0,   def_gain=math.sqrt(2), cuda_idx=9, ref="x", has_2nd_grad=True),
}


def _launch(x, b, xref, yref, dy, grad, dim, act_idx, alpha, gain, clamp):
    _lib.require_gpu(x, b, xref, yref, dy)
    if not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.MgfError("bias_act: x must be non-overlapping and dense (contiguous or channels_last)")
    for t in (xref, yref, dy):
        if t is not None and (t.shape != x.shape or t.stride() != x.stride() or t.dtype != x.dtype):
            raise _lib.MgfError("bias_act: auxiliary tensors must match x in shape, strides and dtype")
    y = torch.empty_like(x)
    step_b, size_b = 1, 1
    if b is not None:
        if b.ndim != 1 or b.dtype != x.dtype or not b.is_contiguous():
            raise _lib.MgfError("bias_act: b must be a contiguous 1-D tensor of x's dtype")
        if not (0 <= dim < x.ndim) or b.shape[0] != x.shape[dim]:
            raise _lib.MgfError("bias_act: b has the wrong number of elements for dim")
        step_b, size_b = x.stride(dim), b.shape[0]
    rc = _lib.lib().mgf_bias_act(y.data_ptr(), x.data_ptr(), _lib.ptr(b), _lib.ptr(xref), _lib.ptr(yref), _lib.ptr(dy),
                                 _lib.dtype_id(x.dtype), x.numel(), step_b, size_b, grad, act_idx, alpha, gain, clamp,
                                 _lib.stream_ptr())
    _lib.check(rc, "bias_act")
    return y


_cache = {}


def _make_op(dim, act, alpha, gain, clamp):
    key = (dim, act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    spec = activation_funcs[act]
    idx = spec.cuda_idx

    class BiasActHip(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            y = x
            if act != "linear" or gain != 1 or clamp >= 0 or b is not None:
                y = _launch(x, b, None, None, None, 0, dim, idx, alpha, gain, clamp)
            ctx.save_for_backward(x if "x" in spec.ref or spec.has_2nd_grad else None,
                                  b if "x" in spec.ref or spec.has_2nd_grad else None,
                                  y if ("y" in spec.ref or clamp >= 0) else None)   # clamp mask needs y (also for linear)
            ctx.has_b = b is not None
            ctx.memory_format = torch.channels_last if (x.ndim == 4 and x.stride(1) == 1 and x.shape[1] > 1) else torch.contiguous_format
            return y

        @staticmethod
        def backward(ctx, dy):
            x, b, y = ctx.saved_tensors
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dx = dy
                if act != "linear" or gain != 1 or clamp >= 0:
                    dx = BiasActHipGrad.apply(dy.contiguous(memory_format=ctx.memory_format), x, b, y)
            if ctx.needs_input_grad[1] and ctx.has_b:
                db = dx.sum([i for i in range(dx.ndim) if i != dim])
            return dx, db

    class BiasActHipGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y):
            dx = _launch(dy, b, x, y, None, 1, dim, idx, alpha, gain, clamp)
            ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx):
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActHipGrad.apply(d_dx.contiguous(), x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = _launch(d_dx.contiguous(), b, x, y, dy, 2, dim, idx, alpha, gain, clamp)
            if spec.has_2nd_grad and ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
            return d_dy, d_x, d_b, None

    _cache[key] = BiasActHip
    return BiasActHip


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None, impl="cuda"):
    """Fused bias + activation + gain + clamp (see the reference docstring, bias_act.py:47-77)."""
    assert isinstance(x, torch.Tensor)
    if impl == "ref":
        raise NotImplementedError("impl='ref' is not part of the MI355X package; use oracle.ops_ref.bias_act_ref in tests")
    assert impl in ("cuda", "hip")
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clamp = float(clamp if clamp is not None else -1)
    _lib.require_gpu(x, b)
    return _make_op(dim, act, alpha, gain, clamp).apply(x, b)
