"""Re-draws one case of tests/test_hip_fuzz.py::test_bias_act_random under MGF_FUZZ_OFFSET and prints where the gradient differs:  MGF_FUZZ_OFFSET=6 python tools/soak_bias_act_probe.py 90"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from morphganformer_amd.torch_utils.ops import bias_act
from oracle.ops_ref import bias_act_ref

OFF = int(os.environ.get("MGF_FUZZ_OFFSET", "0")) * 100003
want = int(sys.argv[1])
acts = ["linear", "relu", "lrelu", "tanh", "sigmoid", "elu", "selu", "softplus", "swish"]
r = np.random.default_rng(303 + OFF)
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
    if case != want:
        continue
    torch.manual_seed(2000 + case + OFF)
    x = torch.randn(*shape).to(dtype)
    b = torch.randn(shape[dim]).to(dtype) if has_b else None
    xr = x.double().requires_grad_(True)
    yr = bias_act_ref(xr, None if b is None else b.double(), dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
    gy = torch.randn(yr.shape, dtype=torch.float64)
    (gr,) = torch.autograd.grad((yr * gy).sum(), xr)
    xg = x.cuda().requires_grad_(True)
    yg = bias_act.bias_act(xg, None if b is None else b.cuda(), dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
    (gx,) = torch.autograd.grad((yg * gy.to(dtype).cuda()).sum(), xg)
    print(shape, dim, act, alpha, gain, clamp, has_b, dtype)
    d = (gx.double().cpu() - gr).abs()
    idx = torch.nonzero(d > 1e-2 * gr.abs().max())
    print("elements off:", len(idx), "of", d.numel(), "max |gr|", float(gr.abs().max()))
    bb = b.double().reshape([-1 if i == dim else 1 for i in range(len(shape))]) if has_b else 0
    for i in idx[:12]:
        t = tuple(int(v) for v in i)
        print(t, "x", float(x[t]), "x+b", float((x.double() + bb)[t]), "y ref", float(yr[t]), "y hip", float(yg[t]), "gy", float(gy[t]), "g ref", float(gr[t]), "g hip", float(gx[t]))
