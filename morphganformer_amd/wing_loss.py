"""Wing loss on landmark tensors -- same contract as the reference's wing_loss.WingLoss (wing_loss.py:13-28):
WingLoss(omega=10, epsilon=2)(pred, target) -> 0-dim float64 tensor (the drivers feed [68,2] DoubleTensors,
1024_example_wing_loss_perceptual_sqz_MSE.py:138-139,169-173).  Runs on the device kernel behind `mgf_wing_loss_f64`."""
from __future__ import annotations

import torch

from . import _lib


class WingLoss(torch.nn.Module):
    def __init__(self, omega=10, epsilon=2):
        super().__init__()
        self.omega = omega
        self.epsilon = epsilon

    def forward(self, pred, target):
        _lib.require_gpu(pred, target)
        p = pred.contiguous().double()
        t = target.contiguous().double()
        assert p.shape == t.shape
        out = torch.empty([], dtype=torch.float64, device=p.device)
        rc = _lib.lib().mgf_wing_loss_f64(out.data_ptr(), p.data_ptr(), t.data_ptr(), 1, p.numel(), float(self.omega),
                                          float(self.epsilon), None, -1, _lib.stream_ptr())
        _lib.check(rc, "wing_loss")
        return out


class AdaptiveWingLoss(torch.nn.Module):
    """adaptive_wing_loss.AdaptiveWingLoss(omega=14, theta=0.5, epsilon=1, alpha=2.1)(pred, target) -> 0-dim float64 tensor
    (adaptive_wing_loss.py:12-39; the heat-map variant of 1024_example_wing_loss_adaptive.py)."""

    def __init__(self, omega=14, theta=0.5, epsilon=1, alpha=2.1):
        super().__init__()
        self.omega, self.theta, self.epsilon, self.alpha = omega, theta, epsilon, alpha

    def forward(self, pred, target):
        _lib.require_gpu(pred, target)
        p = pred.contiguous().double()
        t = target.contiguous().double()
        assert p.shape == t.shape
        out = torch.empty([], dtype=torch.float64, device=p.device)
        rc = _lib.lib().mgf_adaptive_wing_loss_f64(out.data_ptr(), p.data_ptr(), t.data_ptr(), 1, p.numel(), float(self.omega),
                                                   float(self.theta), float(self.epsilon), float(self.alpha), None, -1, _lib.stream_ptr())
        _lib.check(rc, "adaptive_wing_loss")
        return out
