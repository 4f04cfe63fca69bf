"""The FaceNet embedder of the biometric driver on MI355X.

Reference: `model = InceptionResnetV1(pretrained='vggface2').eval()` and, per loop step, `MSE(model(img_gen), model(target))` on the
flattened embeddings, with the generated 1024^2 image handed to the model UN-RESIZED (1024_example_FaceNet_percept.py:30-32,147-158;
`facenet_feature`'s 224-px resize at :34-41 is defined there and never called by the loop).  `facenet_pytorch` is a third-party package
that the reference neither vendors nor pins, and its weights are a remote fetch: nothing of it exists offline.  This module restates the
published topology of facenet_pytorch's `InceptionResnetV1` (models/inception_resnet_v1.py) from its state_dict key names:

    conv2d_1a 3x3/2 (3->32) -> conv2d_2a 3x3 (32) -> conv2d_2b 3x3 pad 1 (64) -> maxpool 3/2 -> conv2d_3b 1x1 (80) -> conv2d_4a 3x3 (192)
    -> conv2d_4b 3x3/2 (256) -> 5 x Block35(scale 0.17) -> Mixed_6a (896) -> 10 x Block17(0.10) -> Mixed_7a (1792) -> 5 x Block8(0.20)
    -> Block8(scale 1, no ReLU) -> AdaptiveAvgPool2d(1) -> [dropout: identity in eval] -> last_linear (1792 -> 512, no bias)
    -> last_bn (BatchNorm1d) -> F.normalize(p=2, dim=1)                                     (classify=False: the embedding)
    BasicConv2d = Conv2d(bias=False) -> BatchNorm2d(eps=1e-3) -> ReLU
    Block35(x)  = relu(conv2d_1x1(cat[b0: 1x1 32 | b1: 1x1 32, 3x3 32 | b2: 1x1 32, 3x3 32, 3x3 32]) * scale + x)        (256 channels)
    Block17(x)  = relu(conv2d_1x1(cat[b0: 1x1 128 | b1: 1x1 128, 1x7 128, 7x1 128]) * scale + x)                         (896 channels)
    Block8(x)   = [relu](conv2d_1x1(cat[b0: 1x1 192 | b1: 1x1 192, 1x3 192, 3x1 192]) * scale + x)                       (1792 channels)
    Mixed_6a    = cat[3x3/2 384 | 1x1 192, 3x3 pad 1 192, 3x3/2 256 | maxpool 3/2]
    Mixed_7a    = cat[1x1 256, 3x3/2 384 | 1x1 256, 3x3/2 256 | 1x1 256, 3x3 pad 1 256, 3x3/2 256 | maxpool 3/2]

Every convolution runs on the library's MFMA kernels: eval-mode BatchNorm is folded into the weights (float64) and the bias port, ReLU
into the epilogue; branch outputs are written straight into their channel slice of the concat buffer; the residual blocks' closing 1x1 is
ONE mgf_conv1x1_f32 launch whose epilogue adds the block input and applies the ReLU behind the add (MGF_ACT_RELU_POST), the block scale
folded into its weights and bias; the 3x3 / stride-1 / pad-1 layers take the Winograd form-3 kernel when the launch fills the chip.
Gradient mode: `keep_activations = True` before the forward pass, then `backward(demb)` (data gradients on the same kernels with
transposed taps, ReLU masks off the stored outputs).  Weights are injectable (`state`: facenet_pytorch's state_dict); offline they are
seeded random (`random_state`).  PARITY IS UNPINNED:
no golden vector of the real package can exist here; the oracle (oracle/embed_ref.py: inception_resnet_v1_ref) restates the same
published topology with torch's own conv ops and the tests compare against it on seeded weights.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from . import conv as cv

BN_EPS = 1e-3
EMB = 512

# BasicConv2d rows: (cin, cout, (kh, kw), stride, (pad_y, pad_x))
STEM = [("conv2d_1a", (3, 32, (3, 3), 2, (0, 0))), ("conv2d_2a", (32, 32, (3, 3), 1, (0, 0))), ("conv2d_2b", (32, 64, (3, 3), 1, (1, 1))),
        ("maxpool_3a", None), ("conv2d_3b", (64, 80, (1, 1), 1, (0, 0))), ("conv2d_4a", (80, 192, (3, 3), 1, (0, 0))),
        ("conv2d_4b", (192, 256, (3, 3), 2, (0, 0)))]
# residual blocks: channels, [branches of BasicConv2d rows], cat width
BLOCK35 = (256, [[(256, 32, (1, 1), 1, (0, 0))],
                 [(256, 32, (1, 1), 1, (0, 0)), (32, 32, (3, 3), 1, (1, 1))],
                 [(256, 32, (1, 1), 1, (0, 0)), (32, 32, (3, 3), 1, (1, 1)), (32, 32, (3, 3), 1, (1, 1))]])
BLOCK17 = (896, [[(896, 128, (1, 1), 1, (0, 0))],
                 [(896, 128, (1, 1), 1, (0, 0)), (128, 128, (1, 7), 1, (0, 3)), (128, 128, (7, 1), 1, (3, 0))]])
BLOCK8 = (1792, [[(1792, 192, (1, 1), 1, (0, 0))],
                 [(1792, 192, (1, 1), 1, (0, 0)), (192, 192, (1, 3), 1, (0, 1)), (192, 192, (3, 1), 1, (1, 0))]])
# reduction blocks: [branches]; the max-pool branch comes last
MIXED_6A = [[(256, 384, (3, 3), 2, (0, 0))],
            [(256, 192, (1, 1), 1, (0, 0)), (192, 192, (3, 3), 1, (1, 1)), (192, 256, (3, 3), 2, (0, 0))]]
MIXED_7A = [[(896, 256, (1, 1), 1, (0, 0)), (256, 384, (3, 3), 2, (0, 0))],
            [(896, 256, (1, 1), 1, (0, 0)), (256, 256, (3, 3), 2, (0, 0))],
            [(896, 256, (1, 1), 1, (0, 0)), (256, 256, (3, 3), 1, (1, 1)), (256, 256, (3, 3), 2, (0, 0))]]
# (prefix, spec, scale, relu) of the residual stages in execution order
RES_STAGES = [("repeat_1", BLOCK35, 5, 0.17), ("repeat_2", BLOCK17, 10, 0.10), ("repeat_3", BLOCK8, 5, 0.20)]


def _branch_key(prefix, bi, li, nlayers):
    """facenet_pytorch names a one-layer branch `branchN` (a BasicConv2d) and a longer one `branchN.L` (an nn.Sequential of them)."""
    return f"{prefix}.branch{bi}" if nlayers == 1 else f"{prefix}.branch{bi}.{li}"


def layer_table():
    """[(state_dict prefix, cin, cout, (kh, kw))] of every BasicConv2d, in definition order (the key names of facenet_pytorch)."""
    rows = [(name, *spec[:3]) for name, spec in STEM if spec is not None]

    def block(prefix, branches):
        for bi, br in enumerate(branches):
            for li, spec in enumerate(br):
                rows.append((_branch_key(prefix, bi, li, len(br)), *spec[:3]))

    for r in range(5):
        block(f"repeat_1.{r}", BLOCK35[1])
    block("mixed_6a", MIXED_6A)
    for r in range(10):
        block(f"repeat_2.{r}", BLOCK17[1])
    block("mixed_7a", MIXED_7A)
    for r in range(5):
        block(f"repeat_3.{r}", BLOCK8[1])
    block("block8", BLOCK8[1])
    return rows


def residual_table():
    """[(prefix, cat width, channels)] of the closing 1x1 `conv2d` (with bias) of every residual block."""
    rows = []
    for prefix, (c, branches), reps, _ in RES_STAGES:
        rows += [(f"{prefix}.{r}", sum(br[-1][1] for br in branches), c) for r in range(reps)]
    rows.append(("block8", sum(br[-1][1] for br in BLOCK8[1]), 1792))
    return rows


def random_state(seed=0):
    """Seeded stand-in weights under facenet_pytorch's state_dict key names (numpy float32): He-scaled convs, BatchNorm with non-trivial
    affine and running statistics."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}

    def bn(name, c):
        sd[name + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[name + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    for name, cin, cout, (kh, kw) in layer_table():
        sd[name + ".conv.weight"] = (rng.standard_normal((cout, cin, kh, kw)) * math.sqrt(2.0 / (cin * kh * kw))).astype(np.float32)
        bn(name + ".bn", cout)
    for name, ccat, c in residual_table():
        sd[name + ".conv2d.weight"] = (rng.standard_normal((c, ccat, 1, 1)) * math.sqrt(1.0 / ccat)).astype(np.float32)
        sd[name + ".conv2d.bias"] = (rng.standard_normal(c) * 0.05).astype(np.float32)
    sd["last_linear.weight"] = (rng.standard_normal((EMB, 1792)) / math.sqrt(1792)).astype(np.float32)
    bn("last_bn", EMB)
    return sd


def conv_gflop(h, w):
    """Algorithmic GFLOP of one forward on an h x w image (2 * taps * cin * cout per output pixel), for the bench's accounting."""
    total, (hh, ww) = 0.0, (h, w)
    out_hw = lambda hh, ww, k, s, p: ((hh + 2 * p[0] - k[0]) // s + 1, (ww + 2 * p[1] - k[1]) // s + 1)
    for name, spec in STEM:
        if spec is None:
            hh, ww = (hh - 3) // 2 + 1, (ww - 3) // 2 + 1
            continue
        cin, cout, k, s, p = spec
        hh, ww = out_hw(hh, ww, k, s, p)
        total += 2.0 * k[0] * k[1] * cin * cout * hh * ww

    def res(c, branches, reps):
        nonlocal total
        per = sum(2.0 * k[0] * k[1] * ci * co for br in branches for (ci, co, k, s, p) in br) + 2.0 * sum(br[-1][1] for br in branches) * c
        total += reps * per * hh * ww

    def mixed(branches):
        nonlocal total, hh, ww
        oh, ow = (hh - 3) // 2 + 1, (ww - 3) // 2 + 1
        for br in branches:
            for (ci, co, k, s, p) in br:
                px = oh * ow if s == 2 else hh * ww
                total += 2.0 * k[0] * k[1] * ci * co * px
        hh, ww = oh, ow

    res(*BLOCK35, 5); mixed(MIXED_6A); res(*BLOCK17, 10); mixed(MIXED_7A); res(*BLOCK8, 6)
    return total / 1e9


class _Conv:
    """One BasicConv2d (BatchNorm folded) or closing 1x1, packed for the kernels."""
    __slots__ = ("pc", "bias", "k", "stride", "pad", "wino", "w_raw", "cin", "cout")


class InceptionResnetV1Embedder:
    """embed_image(img [n,3,H,W] float32 in the generator's value range) -> [n,512] unit-norm embeddings, like `model(img)` of the driver
    (any H, W >= 75; the driver feeds the un-resized 1024^2 image).  Workspaces are preallocated per (n, H, W): graph-capturable."""

    def __init__(self, state=None, n=1, device="cuda", seed=0):
        _lib.lib()
        self.device = torch.device(device)
        self.random_weights = state is None
        sd = state if state is not None else random_state(seed)
        g = lambda k: np.asarray(sd[k].detach().cpu() if isinstance(sd[k], torch.Tensor) else sd[k], dtype=np.float64)
        dev = self.device
        t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)

        def basic(name, spec):
            cin, cout, k, stride, pad = spec
            w = g(name + ".conv.weight")
            assert w.shape == (cout, cin, *k), (name, w.shape)
            s = g(name + ".bn.weight") / np.sqrt(g(name + ".bn.running_var") + BN_EPS)
            L = _Conv()
            wf = t32(w * s[:, None, None, None])
            L.pc, L.bias = cv.pack_weights(wf), t32(g(name + ".bn.bias") - g(name + ".bn.running_mean") * s)
            L.k, L.stride, L.pad, L.cin, L.cout = k, stride, pad, cin, cout
            L.wino = cv.winograd2_weights(wf) if (k == (3, 3) and stride == 1 and pad == (1, 1) and cin % 4 == 0 and cout % 32 == 0) else None
            L.w_raw = wf if cin <= 4 else None              # the 3-channel stem: the streaming kernel takes the torch layout
            return L

        def closing(name, ccat, c, scale):
            L = _Conv()
            L.pc = cv.pack_weights(t32(g(name + ".conv2d.weight") * scale))
            L.bias = t32(g(name + ".conv2d.bias") * scale)
            L.k, L.stride, L.pad, L.cin, L.cout, L.wino, L.w_raw = (1, 1), 1, (0, 0), ccat, c, None, None
            return L

        self.stem = [(name, basic(name, spec) if spec is not None else None) for name, spec in STEM]
        self.stages = []                                     # ("res", channels, [blocks], relu flags) / ("mixed", [branches])

        def res_blocks(prefix, spec, reps, scale, names=None, relu=True):
            c, branches = spec
            blocks = []
            for r in range(reps):
                p = names[r] if names else f"{prefix}.{r}"
                brs = [[basic(_branch_key(p, bi, li, len(br)), row) for li, row in enumerate(br)] for bi, br in enumerate(branches)]
                blocks.append((brs, closing(p, sum(br[-1][1] for br in branches), c, scale), relu))
            return blocks

        mixed = lambda prefix, branches: [[basic(_branch_key(prefix, bi, li, len(br)), row) for li, row in enumerate(br)]
                                          for bi, br in enumerate(branches)]
        self.stages.append(("res", 256, res_blocks("repeat_1", BLOCK35, 5, 0.17)))
        self.stages.append(("mixed", 896, mixed("mixed_6a", MIXED_6A)))
        self.stages.append(("res", 896, res_blocks("repeat_2", BLOCK17, 10, 0.10)))
        self.stages.append(("mixed", 1792, mixed("mixed_7a", MIXED_7A)))
        self.stages.append(("res", 1792, res_blocks("repeat_3", BLOCK8, 5, 0.20) + res_blocks(None, BLOCK8, 1, 1.0, names=["block8"], relu=False)))
        # last_linear (no bias) + last_bn (BatchNorm1d, eval) folded into one affine map
        sb = g("last_bn.weight") / np.sqrt(g("last_bn.running_var") + BN_EPS)
        self.fc_w = t32(g("last_linear.weight") * sb[:, None])
        self.fc_b = t32(g("last_bn.bias") - g("last_bn.running_mean") * sb)
        self.n, self.hw = None, None
        self._n0 = n
        self.keep_activations = False              # gradient mode sets this before the forward pass: backward() reads every layer output
        self.out = torch.empty(n, EMB, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ workspace
    def clone_for(self, n):
        """An instance sharing the packed weights but no mutable workspace (BiometricLoss keeps one for the target images)."""
        other = InceptionResnetV1Embedder.__new__(InceptionResnetV1Embedder)
        other.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("bufs", "out", "n", "hw", "_g", "_kept", "_x_in")})
        other.n, other.hw, other._n0, other.keep_activations = None, None, n, False
        other.out = torch.empty(n, EMB, dtype=torch.float32, device=self.device)
        return other

    def _alloc(self, n, h=None, w=None):
        if h is None:
            if self.hw is None:                              # size not known yet: embed_image allocates on first use
                self._n0 = n
                self.out = torch.empty(n, EMB, dtype=torch.float32, device=self.device)
                return
            h, w = self.hw
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
        keep = self.keep_activations                          # gradient mode: every layer output keeps its own tensor for backward()
        self.n, self.hw, self._kept = n, (h, w), keep
        self._g = None                                        # backward workspace (built on first use)
        B = {"stem": []}
        hh, ww, c = h, w, 3
        out_hw = lambda L, hh, ww: ((hh + 2 * L.pad[0] - L.k[0]) // L.stride + 1, (ww + 2 * L.pad[1] - L.k[1]) // L.stride + 1)
        for name, L in self.stem:
            if L is None:
                hh, ww = (hh - 3) // 2 + 1, (ww - 3) // 2 + 1
            else:
                (hh, ww), c = out_hw(L, hh, ww), L.cout
            if hh < 1 or ww < 1:
                raise _lib.MgfError(f"InceptionResnetV1: a {h}x{w} image is too small")
            B["stem"].append(e(n, c, hh, ww))
        B["stages"] = []

        def inter_bufs(branches, hh, ww, shared):
            """{(branch, layer): tensor} for the non-final layers of every branch: own tensors (keep) or views of two ping-pong buffers."""
            out = {}
            for bi, br in enumerate(branches):
                ih, iw = hh, ww
                for li, L in enumerate(br[:-1]):
                    ih, iw = out_hw(L, ih, iw)
                    out[(bi, li)] = e(n, L.cout, ih, iw) if shared is None else shared[li & 1][:n * L.cout * ih * iw].view(n, L.cout, ih, iw)
            return out

        for kind, c, body in self.stages:
            if kind == "res":
                ccat = body[0][1].cin
                tmp = max(max(L.cout for L in br[:-1]) if len(br) > 1 else 0 for br in body[0][0])
                shared_t = None if keep else [e(n * tmp * hh * ww), e(n * tmp * hh * ww)]
                shared_cat, shared_x = (None, None) if keep else (e(n, ccat, hh, ww), [e(n, c, hh, ww), e(n, c, hh, ww)])
                blocks = [dict(cat=e(n, ccat, hh, ww) if keep else shared_cat, t=inter_bufs(brs, hh, ww, shared_t),
                               x=e(n, c, hh, ww) if keep else shared_x[bi & 1]) for bi, (brs, _, _) in enumerate(body)]
                B["stages"].append(dict(blocks=blocks, hw=(hh, ww)))
            else:
                oh, ow = (hh - 3) // 2 + 1, (ww - 3) // 2 + 1
                if oh < 1 or ow < 1:
                    raise _lib.MgfError(f"InceptionResnetV1: a {h}x{w} image is too small")
                tmp = max(L.cout for br in body for L in br[:-1])
                shared_t = None if keep else [e(n * tmp * hh * ww), e(n * tmp * hh * ww)]
                B["stages"].append(dict(cat=e(n, c, oh, ow), t=inter_bufs(body, hh, ww, shared_t), pool=e(n, body[0][0].cin, oh, ow), hw=(hh, ww)))
                hh, ww = oh, ow
        B["mean"] = e(n, 1792)
        B["pre"] = e(n, EMB)
        self.bufs = B
        if self.out.shape[0] != n:
            self.out = e(n, EMB)

    # ------------------------------------------------------------------ forward
    def _conv(self, x, L, out, choff=0, act="relu", residual=None):
        ep = _lib.make_epilogue(bias=L.bias, act=act, residual=residual)
        n, _, h, w = x.shape
        if L.w_raw is not None and L.stride == 2 and L.k == (3, 3) and L.pad == (0, 0) and choff == 0 and act == "relu":
            return cv.conv3x3s2_few_inputs(x, L.w_raw, bias=L.bias, relu=True, out=out)
        if L.wino is not None and cv.winograd_fills_chip(n, L.cout, h, w):
            return cv.winograd2_forward(x, L.wino, epilogue=ep, out=out, out_choff=choff)
        return cv.conv_forward(x, L.pc, stride=L.stride, pad=L.pad, epilogue=ep, out=out, out_choff=choff)

    def _pool(self, y, x):
        n, c, h, w = x.shape
        _lib.check(_lib.lib().mgf_maxpool_s2_floor_f32(y.data_ptr(), x.data_ptr(), n * c, h, w, 3, _lib.stream_ptr()), "maxpool")
        return y

    def _branch(self, x, br, bi, inter, cat, choff):
        """A chain of BasicConv2d: intermediate results go to `inter[(branch, layer)]`, the last layer writes its slice of the concat buffer."""
        h = x
        for li, L in enumerate(br):
            if li == len(br) - 1:
                self._conv(h, L, cat, choff)
            else:
                h = self._conv(h, L, inter[(bi, li)])
        return br[-1].cout

    def embed_image(self, img, out=None):
        _lib.require_gpu(img, out)
        n, c, h, w = img.shape
        assert c == 3 and img.dtype == torch.float32
        if (self.n, self.hw, getattr(self, "_kept", None)) != (n, (h, w), self.keep_activations):
            self._alloc(n, h, w)
        B = self.bufs
        x = self._x_in = img.contiguous()
        for (name, L), buf in zip(self.stem, B["stem"]):
            x = self._pool(buf, x) if L is None else self._conv(x, L, buf)
        for (kind, c, body), S in zip(self.stages, B["stages"]):
            if kind == "res":
                for (brs, close, relu), K in zip(body, S["blocks"]):
                    off = 0
                    for bi, br in enumerate(brs):
                        off += self._branch(x, br, bi, K["t"], K["cat"], off)
                    # out = [relu](conv2d(cat) * scale + x): scale folded into the weights / bias, the add and the ReLU in the epilogue
                    x = self._conv(K["cat"], close, K["x"], act="relu_post" if relu else "linear", residual=x)
            else:
                off = 0
                for bi, br in enumerate(body):
                    off += self._branch(x, br, bi, S["t"], S["cat"], off)
                self._pool(S["pool"], x)
                S["cat"][:, off:].copy_(S["pool"])                          # the max-pool branch: last slice of the concat buffer
                x = S["cat"]
        L_, st = _lib.lib(), _lib.stream_ptr()
        _lib.check(L_.mgf_spatial_mean_f32(B["mean"].data_ptr(), x.data_ptr(), n * 1792, x.shape[2] * x.shape[3], st), "spatial_mean")
        for r0 in range(0, n, 16):                                          # the GEMV kernel takes at most 16 rows per launch
            rows = min(16, n - r0)
            _lib.check(L_.mgf_linear_f32(B["pre"][r0:].data_ptr(), B["mean"][r0:].data_ptr(), self.fc_w.data_ptr(), self.fc_b.data_ptr(), rows,
                                         1792, EMB, st), "linear")
        out = self.out if out is None else out
        _lib.check(L_.mgf_l2_normalize_f32(out.data_ptr(), B["pre"].data_ptr(), n, EMB, 1e-12, st), "l2_normalize")
        return out

    # ------------------------------------------------------------------ backward (gradient mode)
    def _gpack(self, L):
        """Transposed taps of a layer's data gradient, built once: stride 1 -> the flipped, channel-transposed kernel (a correlation's
        gradient is a true convolution), stride 2 -> the same taps for the transposed-conv kernel."""
        gp = getattr(self, "_gpacks", None)
        if gp is None:
            gp = self._gpacks = {}
        t = gp.get(id(L))
        if t is None:
            t = gp[id(L)] = cv.transpose_packed(L.pc, flip=(L.stride == 1 and L.k != (1, 1)))
        return t

    def _gbuf(self, key, shape):
        t = self._g.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._g[key] = torch.empty(shape, dtype=torch.float32, device=self.device)
        return t

    def _mask(self, key, dy, y, c, dy_off=0, y_off=0):
        """dx = dy[:, dy_off : dy_off + c] where y[:, y_off : y_off + c] > 0 else 0 (ReLU backward off the layer OUTPUT), dense."""
        n, hw = dy.shape[0], dy.shape[2] * dy.shape[3]
        dx = self._gbuf(key, (n, c, dy.shape[2], dy.shape[3]))
        _lib.check(_lib.lib().mgf_relu_bwd_slice_f32(dx.data_ptr(), dy.data_ptr(), dy.shape[1], dy_off, y.data_ptr(), y.shape[1], y_off, n, c, hw,
                                                     _lib.stream_ptr()), "relu_bwd_slice")
        return dx

    def _dgrad(self, key, g, L, in_shape, accumulate_into=None):
        """d(input) of one (already ReLU-masked) BasicConv2d / closing conv: g [n, cout, oh, ow] -> [n, cin, H, W].  accumulate_into: add the
        result to that tensor in place (stride-1 layers: through the conv epilogue's residual port) and return it."""
        n, cin, H, W = in_shape
        T = self._gpack(L)
        if L.stride == 1:
            pad = (L.k[0] - 1 - L.pad[0], L.k[1] - 1 - L.pad[1])
            if accumulate_into is not None:
                return cv.conv_forward(g, T, pad=pad, epilogue=_lib.make_epilogue(residual=accumulate_into), out=accumulate_into)
            return cv.conv_forward(g, T, pad=pad, out=self._gbuf(key, in_shape))
        # 3x3 / stride 2 / no padding: x[2i + k] += w[k] g[i] = the transposed-conv kernel; rows / columns past 2 oh (an even input side) get nothing
        oh, ow = g.shape[2:]
        t = cv.tconv3x3s2_forward(g.contiguous(), T, out=self._gbuf(key + ("t",), (n, cin, 2 * oh + 1, cv.tconv_pitch(ow))))
        if accumulate_into is not None:
            accumulate_into[:, :, :2 * oh + 1, :2 * ow + 1].add_(t)
            return accumulate_into
        dx = self._gbuf(key, in_shape)
        if (H, W) != (2 * oh + 1, 2 * ow + 1):
            dx.zero_()
        dx[:, :, :2 * oh + 1, :2 * ow + 1].copy_(t)
        return dx

    def _branch_bwd(self, key, g, br, bi, inter, x_in, dxin):
        """Backward of one branch: g = the masked gradient of its LAST layer's output; intermediate outputs from `inter`; the first layer's
        input is the block input x_in, whose gradient accumulates into dxin."""
        for li in range(len(br) - 1, -1, -1):
            L = br[li]
            if li == 0:
                self._dgrad(key + (bi, li), g, L, tuple(x_in.shape), accumulate_into=dxin)
            else:
                y_prev = inter[(bi, li - 1)]
                gp = self._dgrad(key + (bi, li), g, L, tuple(y_prev.shape))
                g = self._mask(key + (bi, li, "m"), gp, y_prev, y_prev.shape[1])

    def backward(self, demb, dimg=None, accumulate=False):
        """demb [n, 512] -> the gradient with respect to the image of the latest embed_image() call (keep_activations must have been set
        before that call).  With `dimg` the result is written (added, when `accumulate`) there; otherwise a [n, 3, H, W] tensor is returned."""
        _lib.require_gpu(demb, dimg)
        if not getattr(self, "_kept", False):
            raise _lib.MgfError("InceptionResnetV1Embedder.backward: set keep_activations = True before the forward pass (gradient mode)")
        if self._g is None:
            self._g = {}
        L_, st, n, B = _lib.lib(), _lib.stream_ptr(), self.n, self.bufs
        dpre = self._gbuf(("dpre",), (n, EMB))
        _lib.check(L_.mgf_l2_normalize_bwd_f32(dpre.data_ptr(), demb.contiguous().data_ptr(), B["pre"].data_ptr(), n, EMB, 1e-12, st), "l2_normalize_bwd")
        dmean = self._gbuf(("dmean",), (n, 1792))
        for r0 in range(0, n, 16):
            rows = min(16, n - r0)
            _lib.check(L_.mgf_linear_bwd_f32(dmean[r0:].data_ptr(), dpre[r0:].data_ptr(), self.fc_w.data_ptr(), rows, 1792, EMB, st), "linear_bwd")
        last = B["stages"][-1]["blocks"][-1]["x"]
        dx = self._gbuf(("dx", len(self.stages), 0), tuple(last.shape))
        _lib.check(L_.mgf_spatial_mean_bwd_f32(dx.data_ptr(), dmean.data_ptr(), n * 1792, last.shape[2] * last.shape[3], st), "spatial_mean_bwd")
        # the input of stage i: the output of stage i - 1 (the last block's x, or the mixed stage's concat buffer), the stem's output for i = 0
        stage_in = [B["stem"][-1]] + [(S["blocks"][-1]["x"] if "blocks" in S else S["cat"]) for S in B["stages"][:-1]]
        for si in range(len(self.stages) - 1, -1, -1):
            (kind, c, body), S = self.stages[si], B["stages"][si]
            if kind == "res":
                for bi in range(len(body) - 1, -1, -1):
                    (brs, close, relu), K = body[bi], S["blocks"][bi]
                    x_in = S["blocks"][bi - 1]["x"] if bi > 0 else stage_in[si]
                    # out = [relu](conv2d(cat) + x): d(pre) = dx masked by the block OUTPUT; it is also the residual path's d(x_in)
                    dxin = self._mask(("dx", si, bi), dx, K["x"], c) if relu else dx
                    # (the closing conv's gradient needs d(pre) BEFORE the branches add to it: keep a copy)
                    dpre_blk = self._gbuf(("dpre_blk", si), tuple(dxin.shape))
                    dpre_blk.copy_(dxin)
                    off = 0
                    for b, br in enumerate(brs):
                        cb = br[-1].cout
                        gcat = cv.conv_forward(dpre_blk, self._closing_slice(close, off, cb), out=self._gbuf(("gcat", si, b), (n, cb, *K["cat"].shape[2:])))
                        g = self._mask(("gm", si, b), gcat, K["cat"], cb, 0, off)
                        self._branch_bwd(("br", si), g, br, b, K["t"], x_in, dxin)
                        off += cb
                    dx = dxin
            else:
                x_in = stage_in[si]
                dxin = self._gbuf(("dx", si, 0), tuple(x_in.shape))
                off = sum(br[-1].cout for br in body)
                cpool = x_in.shape[1]
                gpool = self._gbuf(("gpool", si), tuple(S["pool"].shape))
                gpool.copy_(dx[:, off:off + cpool])
                _lib.check(L_.mgf_maxpool_s2_floor_bwd_f32(dxin.data_ptr(), gpool.data_ptr(), x_in.data_ptr(), n * cpool, x_in.shape[2], x_in.shape[3],
                                                           3, st), "maxpool_bwd")
                off = 0
                for b, br in enumerate(body):
                    cb = br[-1].cout
                    g = self._mask(("gm", si, b), dx, S["cat"], cb, off, off)
                    self._branch_bwd(("br", si), g, br, b, S["t"], x_in, dxin)
                    off += cb
                dx = dxin
        # stem, in reverse: every BasicConv2d output is a ReLU output
        stem_in = [self._x_in] + B["stem"][:-1]
        for i in range(len(self.stem) - 1, -1, -1):
            (name, L), y, x_in = self.stem[i], B["stem"][i], stem_in[i]
            if L is None:
                dxi = self._gbuf(("stem", i), tuple(x_in.shape))
                _lib.check(L_.mgf_maxpool_s2_floor_bwd_f32(dxi.data_ptr(), dx.contiguous().data_ptr(), x_in.data_ptr(), n * x_in.shape[1], x_in.shape[2],
                                                           x_in.shape[3], 3, st), "maxpool_bwd")
            else:
                g = self._mask(("stem", i, "m"), dx, y, L.cout)
                dxi = self._dgrad(("stem", i), g, L, tuple(x_in.shape))
            dx = dxi
        if dimg is None:
            return dx
        if accumulate:
            dimg.add_(dx)
        else:
            dimg.copy_(dx)
        return dimg

    def _closing_slice(self, close, off, cb):
        """The data-gradient weights of a residual block's closing 1x1 for ONE branch of its concat input: out channels [off, off + cb)
        of the transposed kernel, as a packed 1x1 conv c -> cb (so that each branch's gradient leaves as a dense tensor)."""
        cache = getattr(self, "_cslices", None)
        if cache is None:
            cache = self._cslices = {}
        key = (id(close), off, cb)
        t = cache.get(key)
        if t is None:
            w = close.pc.wp[0, off:off + cb, :close.cout]                     # packed [tap = 1][cin = ccat][cout_pad]: rows = cat channels
            t = cache[key] = cv.pack_weights(w.reshape(cb, close.cout, 1, 1).contiguous())
        return t

    __call__ = embed_image
