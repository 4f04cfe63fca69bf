"""fma(a, b, c) = a * b + c  (reference: torch_utils/ops/fma.py:7-37).  In the reference this only exists to give
addcmul a cheaper custom backward during GAN training; the projection path calls it (if at all) in inference, where it is a
plain fused multiply-add, so it maps to torch.addcmul on the device."""
import torch


def fma(a, b, c):
    return torch.addcmul(c, a, b)
