"""The biometric term of the projection objective: an IResNet face embedder on MI355X and the
embedding-MSE loss the drivers build from it.

Reference: the wired variant is `facenet_pytorch.InceptionResnetV1` (1024_example_FaceNet_percept.py:147-158: the loss is
`MSE(model(img_gen), model(target))` on flattened embeddings) -- a third-party package that is not vendored and whose weights
are a remote fetch.  The network the reference DOES vendor for this role is `backbones/iresnet.py` (IResNet-18/34/50/100,
ArcFace layout: 112x112 input, 512-d embedding); that is what is built here, from its state_dict key names:

    conv1 3x3 (3->64) -> bn1 -> PReLU -> layer1..4 of IBasicBlocks -> bn2 -> flatten -> fc (512*7*7 -> 512) -> features (BN1d)
    IBasicBlock(x) = bn3(conv2_stride(PReLU(bn2(conv1(bn1(x)))))) + [downsample(x) | x]          (iresnet.py:28-60, 138-160)

Eval-mode BatchNorm is an affine map: after a conv it rides on conv_taps' out_scale/bias ports, the shortcut on its residual
port, `features` is folded into the fc weights in float64; the BatchNorm in FRONT of a zero-padded conv (bn1 of every block)
cannot be folded exactly at the border, so it is an element-wise pass (mgf_channel_affine_prelu_f32), as is PReLU.
No checkpoint exists offline: weights are injectable (`state`), seeded random by default (`random_state`).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from . import conv as cv

LAYERS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 14, 3], 100: [3, 13, 30, 3], 200: [6, 26, 60, 6]}
PLANES = [64, 128, 256, 512]
EPS = 1e-5


def block_table(depth):
    """[(prefix, inplanes, planes, stride, has_downsample)] in execution order (iresnet.py:118-136)."""
    rows, inpl = [], 64
    for li, (planes, blocks) in enumerate(zip(PLANES, LAYERS[depth])):
        for bi in range(blocks):
            stride = 2 if bi == 0 else 1
            rows.append((f"layer{li + 1}.{bi}", inpl, planes, stride, bi == 0))
            inpl = planes
    return rows


def random_state(depth=50, seed=0):
    """Seeded stand-in weights under the reference's state_dict key names (numpy float32): He-scaled convs, BatchNorm with
    non-trivial affine and running statistics, PReLU slopes in (0.1, 0.4)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = (rng.standard_normal((co, ci, k, k)) * math.sqrt(2.0 / (ci * k * k))).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[name + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    conv("conv1", 64, 3, 3); bn("bn1", 64)
    sd["prelu.weight"] = rng.uniform(0.1, 0.4, 64).astype(np.float32)
    for p, inpl, planes, stride, ds in block_table(depth):
        bn(p + ".bn1", inpl); conv(p + ".conv1", planes, inpl, 3); bn(p + ".bn2", planes)
        sd[p + ".prelu.weight"] = rng.uniform(0.1, 0.4, planes).astype(np.float32)
        conv(p + ".conv2", planes, planes, 3); bn(p + ".bn3", planes)
        if ds:
            conv(p + ".downsample.0", planes, inpl, 1); bn(p + ".downsample.1", planes)
    bn("bn2", 512)
    sd["fc.weight"] = (rng.standard_normal((512, 512 * 49)) / math.sqrt(512 * 49)).astype(np.float32)
    sd["fc.bias"] = (rng.standard_normal(512) * 0.05).astype(np.float32)
    bn("features", 512)
    return sd


def _bn_affine(sd, name):
    g = lambda k: np.asarray(sd[f"{name}.{k}"], dtype=np.float64)
    s = g("weight") / np.sqrt(g("running_var") + EPS)
    return s, g("bias") - g("running_mean") * s


class IResNetEmbedder:
    """embed(x112 [n,3,112,112] in [-1,1]) -> [n,512]; `embed_image` first resizes any [n,3,H,W] image bilinearly to 112x112."""

    def __init__(self, state=None, depth=50, n=1, device="cuda", seed=0):
        _lib.lib()
        self.device = torch.device(device)
        self.depth, self.n = depth, n
        sd = state if state is not None else random_state(depth, seed)
        dev = self.device
        t32 = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a), dtype=np.float32), device=dev)
        pack = lambda k: cv.pack_weights(t32(sd[k]))
        aff = lambda name: tuple(t32(v) for v in _bn_affine(sd, name))
        self.stem = (pack("conv1.weight"), *aff("bn1"), t32(sd["prelu.weight"]))
        self.blocks = []
        for p, inpl, planes, stride, ds in block_table(depth):
            self.blocks.append(dict(
                bn1=aff(p + ".bn1"), conv1=pack(p + ".conv1.weight"), bn2=aff(p + ".bn2"), slope=t32(sd[p + ".prelu.weight"]),
                conv2=pack(p + ".conv2.weight"), bn3=aff(p + ".bn3"), stride=stride, planes=planes,
                down=(pack(p + ".downsample.0.weight"), *aff(p + ".downsample.1")) if ds else None))
        self.bn2 = aff("bn2")
        sf, tf = _bn_affine(sd, "features")
        wf = np.asarray(sd["fc.weight"], dtype=np.float64) * sf[:, None]
        self.fc_w, self.fc_b = t32(wf), t32(np.asarray(sd["fc.bias"], dtype=np.float64) * sf + tf)
        self._alloc(n)

    def clone_for(self, n):
        """An instance sharing the packed weights but no mutable workspace (BiometricLoss keeps one for the target images)."""
        other = IResNetEmbedder.__new__(IResNetEmbedder)
        other.__dict__.update({k: v for k, v in self.__dict__.items()
                               if k not in ("bufs", "x112", "stem_out", "flat", "out") and not k.startswith("_g")})
        other._alloc(n)
        return other

    def _alloc(self, n):
        self.n = n
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
        self.x112 = e(n, 3, 112, 112)
        self.stem_out = e(n, 64, 112, 112)
        self.bufs = []
        res = 112
        for b in self.blocks:
            inpl = self.blocks[len(self.bufs) - 1]["planes"] if self.bufs else 64
            ores = res // b["stride"]
            self.bufs.append(dict(a=e(n, inpl, res, res), h=e(n, b["planes"], res, res), out=e(n, b["planes"], ores, ores),
                                  idn=e(n, b["planes"], ores, ores) if b["down"] is not None else None))
            res = ores
        assert res == 7
        self.flat = e(n, 512, 7, 7)
        self.out = e(n, 512)

    def _affine(self, y, x, scale=None, shift=None, slope=None):
        n, c = x.shape[:2]
        _lib.check(_lib.lib().mgf_channel_affine_prelu_f32(y.data_ptr(), x.data_ptr(), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(slope),
                                                           n, c, x.shape[2] * x.shape[3], _lib.stream_ptr()), "channel_affine_prelu")
        return y

    def embed(self, x112, out=None):
        _lib.require_gpu(x112, out)
        n = x112.shape[0]
        if n != self.n:
            self._alloc(n)
        assert tuple(x112.shape) == (n, 3, 112, 112) and x112.dtype == torch.float32 and x112.is_contiguous()
        pc, s, t, slope = self.stem
        x = cv.conv_forward(x112, pc, pad=(1, 1), out_scale=s, epilogue=_lib.make_epilogue(bias=t), out=self.stem_out)
        x = self._affine(x, x, slope=slope)
        for b, B in zip(self.blocks, self.bufs):
            a = self._affine(B["a"], x, *b["bn1"])
            h = cv.conv_forward(a, b["conv1"], pad=(1, 1), out_scale=b["bn2"][0], epilogue=_lib.make_epilogue(bias=b["bn2"][1]), out=B["h"])
            h = self._affine(h, h, slope=b["slope"])
            if b["down"] is not None:
                pd, sd_, td = b["down"]
                idn = cv.conv_forward(x, pd, stride=b["stride"], out_scale=sd_, epilogue=_lib.make_epilogue(bias=td), out=B["idn"])
            else:
                idn = x
            x = cv.conv_forward(h, b["conv2"], stride=b["stride"], pad=(1, 1), out_scale=b["bn3"][0],
                                epilogue=_lib.make_epilogue(bias=b["bn3"][1], residual=idn), out=B["out"])
        f = self._affine(self.flat, x, *self.bn2)
        out = self.out if out is None else out
        for r0 in range(0, n, 16):                                   # the GEMV kernel takes at most 16 rows per launch
            rows = min(16, n - r0)
            _lib.check(_lib.lib().mgf_linear_f32(out[r0:].data_ptr(), f[r0:].data_ptr(), self.fc_w.data_ptr(), self.fc_b.data_ptr(), rows,
                                                 512 * 49, 512, _lib.stream_ptr()), "linear")
        return out

    # ------------------------------------------------------------------ gradient mode
    def _grad_ws(self):
        """Transposed taps + workspace of `backward` (built on first use).  BatchNorm scales ride on the dgrad convs' input-scale
        port, which is per sample: they are expanded to [n, c] once."""
        n = self.n
        if getattr(self, "_gp", None) is None:
            self._gp = {"stem": cv.transpose_packed(self.stem[0], flip=True), "blocks": []}
            for b in self.blocks:
                if bool((b["slope"] <= 0).any()):
                    raise _lib.MgfError("IResNet backward: PReLU slopes must be positive (the pre-activation sign is read off the output)")
                self._gp["blocks"].append(dict(
                    c1=cv.transpose_packed(b["conv1"], flip=True),
                    c2=cv.transpose_packed(b["conv2"], flip=(b["stride"] == 1)),      # stride 2: gradient = transposed conv, same taps
                    down=cv.transpose_packed(b["down"][0], flip=False) if b["down"] is not None else None))
            if bool((self.stem[3] <= 0).any()):
                raise _lib.MgfError("IResNet backward: PReLU slopes must be positive")
        if getattr(self, "_gn", None) != n:
            self._gn = n
            ex = lambda t: t.reshape(1, -1).expand(n, -1).contiguous()
            self._gs = dict(stem=ex(self.stem[1]), blocks=[dict(bn2=ex(b["bn2"][0]), bn3=ex(b["bn3"][0]),
                                                                down=ex(b["down"][1]) if b["down"] is not None else None)
                                                           for b in self.blocks])
            e = lambda t: torch.empty_like(t)
            self._gb = [dict(dh=e(B["h"]), dpre=e(B["h"]), da=e(B["a"]), dx=e(B["a"]),
                             t=torch.empty([n, B["h"].shape[1], B["h"].shape[2] + 1, cv.tconv_pitch(B["out"].shape[3])], dtype=torch.float32,
                                           device=self.device) if b["stride"] == 2 else None,
                             dlow=torch.empty([n, B["a"].shape[1], B["out"].shape[2], B["out"].shape[3]], dtype=torch.float32,
                                              device=self.device) if b["down"] is not None else None)
                        for b, B in zip(self.blocks, self.bufs)]
            self._gflat, self._gtop = e(self.flat), e(self.flat)
            self._gstem, self._gx112 = e(self.stem_out), e(self.x112)

    def backward(self, demb, dimg=None, accumulate=False):
        """demb [n,512] -> gradient wrt the image of the latest embed_image()/embed() call.  With `dimg` [n,3,H,W] the result is
        scattered through the bilinear resize into it (added when `accumulate`); otherwise the [n,3,112,112] gradient is returned."""
        _lib.require_gpu(demb, dimg)
        self._grad_ws()
        L, st, n = _lib.lib(), _lib.stream_ptr(), self.n
        for r0 in range(0, n, 16):
            rows = min(16, n - r0)
            _lib.check(L.mgf_linear_bwd_f32(self._gflat[r0:].data_ptr(), demb[r0:].data_ptr(), self.fc_w.data_ptr(), rows, 512 * 49, 512, st),
                       "linear_bwd")
        dx = self._affine(self._gtop, self._gflat, scale=self.bn2[0])
        for b, B, gp, gs, gb in reversed(list(zip(self.blocks, self.bufs, self._gp["blocks"], self._gs["blocks"], self._gb))):
            dxo = dx
            if b["stride"] == 1:
                dh = cv.conv_forward(dxo, gp["c2"], pad=(1, 1), in_scale=gs["bn3"], out=gb["dh"])
            else:
                t = cv.tconv3x3s2_forward(dxo, gp["c2"], in_scale=gs["bn3"], out=gb["t"])      # T[q + 1] = d h[q]: drop row/column 0
                gb["dh"].copy_(t[:, :, 1:, 1:])
                dh = gb["dh"]
            c, hw = dh.shape[1], dh.shape[2] * dh.shape[3]
            _lib.check(L.mgf_prelu_bwd_f32(gb["dpre"].data_ptr(), dh.data_ptr(), B["h"].data_ptr(), b["slope"].data_ptr(), n, c, hw, st),
                       "prelu_bwd")
            da = cv.conv_forward(gb["dpre"], gp["c1"], pad=(1, 1), in_scale=gs["bn2"], out=gb["da"])
            dxi = self._affine(gb["dx"], da, scale=b["bn1"][0])
            if b["down"] is not None:
                low = cv.conv_forward(dxo, gp["down"], in_scale=gs["down"], out=gb["dlow"])
                dxi[:, :, ::b["stride"], ::b["stride"]] += low
            else:
                dxi += dxo
            dx = dxi
        c, hw = dx.shape[1], dx.shape[2] * dx.shape[3]
        _lib.check(L.mgf_prelu_bwd_f32(self._gstem.data_ptr(), dx.data_ptr(), self.stem_out.data_ptr(), self.stem[3].data_ptr(), n, c, hw, st),
                   "prelu_bwd")
        d112 = cv.conv_forward(self._gstem, self._gp["stem"], pad=(1, 1), in_scale=self._gs["stem"], out=self._gx112)
        if dimg is None:
            return d112
        if not accumulate:
            dimg.zero_()
        h, w = dimg.shape[2:]
        if (h, w) == (112, 112):
            dimg += d112
        else:
            _lib.check(L.mgf_resize_bilinear_bwd_f32(dimg.data_ptr(), d112.data_ptr(), n * 3, h, w, 112, 112, st), "resize_bilinear_bwd")
        return dimg

    def embed_image(self, img, out=None):
        """img [n,3,H,W] in [-1,1] -> embedding; bilinear resize (align_corners=False) to the 112x112 ArcFace input."""
        _lib.require_gpu(img)
        n, c, h, w = img.shape
        if n != self.n:
            self._alloc(n)
        if (h, w) == (112, 112):
            return self.embed(img.contiguous(), out)
        _lib.check(_lib.lib().mgf_resize_bilinear_f32(self.x112.data_ptr(), img.contiguous().data_ptr(), n * c, h, w, 112, 112,
                                                      _lib.stream_ptr()), "resize_bilinear")
        return self.embed(self.x112, out)

    __call__ = embed_image


class BiometricLoss:
    """loss[i] = MSE(embed(pred[i]), embed(target)) like `MSE(img_gen_fea, img_fea)` (1024_example_FaceNet_percept.py:147-158);
    the target embedding is computed once per target (the reference recomputes it every step)."""

    def __init__(self, embedder="iresnet50", **kw):
        """embedder: an embedder object (IResNetEmbedder, facenet.InceptionResnetV1Embedder) or a name -- "facenet": the network the
        driver scores with, on the un-resized image (1024_example_FaceNet_percept.py:30-32,147-158); "iresnet18/34/50/100": the vendored
        ArcFace network on a 112x112 resize (backbones/iresnet.py).  kw (state=, n=, device=, seed=) go to the embedder's constructor."""
        if isinstance(embedder, str):
            if embedder == "facenet":
                from .facenet import InceptionResnetV1Embedder
                embedder = InceptionResnetV1Embedder(**kw)
            elif embedder.startswith("iresnet"):
                embedder = IResNetEmbedder(depth=int(embedder[len("iresnet"):]), **kw)
            else:
                raise ValueError(f"unknown embedder {embedder!r} (facenet, iresnet18/34/50/100)")
        self.embedder = embedder
        self._target = None
        self._target_stride = 0
        self._tgt_net, self._tgt_nt = None, None
        self._scratch = None

    def set_target(self, target):
        """One target [1,3,H,W] shared by every candidate, or B targets that pair up with B candidates (lockstep projections)."""
        nt = int(target.shape[0])
        if self._tgt_net is None or self._tgt_nt != nt:           # an instance sharing nothing mutable with the candidates' workspace
            self._tgt_net, self._tgt_nt = self.embedder.clone_for(nt), nt
        emb = self._tgt_net.embed_image(target.float())
        old = getattr(self, "_target", None)
        if old is not None and old.shape == emb.shape:
            old.copy_(emb)            # in place: a captured hipGraph of the projection engine keeps reading this buffer (ProjectionEngine.retarget)
        else:
            self._target = emb.clone()
        self._target_stride = 512 if nt > 1 else 0

    def distance_into(self, out, pred, scale=1.0, accumulate=False):
        """out[i] (+)= scale * mean((embed(pred[i]) - embed(target))^2);  out: float32 [n]."""
        assert self._target is not None, "call set_target first"
        n = pred.shape[0]
        assert self._target_stride == 0 or self._target.shape[0] == n, "B targets pair up with B candidates"
        emb = self.embedder.embed_image(pred)
        need = n * int(_lib.lib().mgf_reduce_scratch_floats())
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.float32, device=emb.device)
        _lib.check(_lib.lib().mgf_mse_f32(out.data_ptr(), emb.data_ptr(), self._target.data_ptr(), n, 512, self._target_stride, float(scale),
                                          int(accumulate), self._scratch.data_ptr(), _lib.stream_ptr()), "mse(embedding)")
        return out

    def grad_into(self, dimg, scale=1.0, accumulate=False):
        """dimg (+)= d(scale * distance)/d(pred) for the pred of the latest distance_into call (gradient mode)."""
        e = self.embedder
        n = e.n
        if getattr(self, "_demb", None) is None or self._demb.shape[0] != n:
            self._demb = torch.empty(n, 512, dtype=torch.float32, device=dimg.device)
        _lib.check(_lib.lib().mgf_mse_grad_f32(self._demb.data_ptr(), e.out.data_ptr(), self._target.data_ptr(), n, 512, self._target_stride, float(scale), 0,
                                               _lib.stream_ptr()), "mse_grad(embedding)")
        return e.backward(self._demb, dimg, accumulate)

    def __call__(self, pred, target):
        self.set_target(target)
        out = torch.zeros(pred.shape[0], dtype=torch.float32, device=pred.device)
        return self.distance_into(out, pred)
