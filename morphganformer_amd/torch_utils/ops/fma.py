"""fma(a, b, c) = a * b + c with broadcasting (reference: torch_utils/ops/fma.py:7-60).

The reference has no plugin behind this name either: its forward is `torch.addcmul` and the point of the op is the custom backward --
each operand's gradient is the product with the OTHER factor summed back to that operand's own shape, computed only for the operands
that need one (cheaper than addcmul's autograd during GAN training, where it sits inside `modulated_conv2d`'s non-fused branch,
networks.py:311-318).  The projection path reaches it at most in inference; it is kept with the reference's semantics, gradients of any
order included (the backward is written in differentiable torch ops), so that `training.networks` source embedded in a checkpoint finds
the name it imports."""
import torch


def _sum_to_shape(t, shape):
    """Undo broadcasting: sum `t` over the dimensions that `shape` does not have or has with extent 1."""
    lead = t.ndim - len(shape)
    assert lead >= 0
    dims = [i for i in range(t.ndim) if t.shape[i] > 1 and (i < lead or shape[i - lead] == 1)]
    if dims:
        t = t.sum(dim=dims, keepdim=True)
    if lead:
        t = t.reshape(t.shape[lead:])
    assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
    return t


class _Fma(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        da = _sum_to_shape(dy * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _sum_to_shape(dy * a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _sum_to_shape(dy, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc


def fma(a, b, c):
    return _Fma.apply(a, b, c)
