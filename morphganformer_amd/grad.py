"""Gradient mode: d(loss)/d(latent) through the HIP generator -- the backward pass torch autograd runs for the reference when a
projection driver differentiates its loss with respect to the latent (the north-star reading of the loop, SURVEY.md section 8a
row P0; the literal reference loop severs this gradient, section 0.1, so gradient mode is an extension whose oracle is autograd
through oracle/generator_ref.py, itself pinned on the reference module's own autograd by the `grad_z` vector of tests/golden/gen_tiny.npz).

    gg = GeneratorGrad(G)
    img = gg.forward(z, noise_mode="const")            # G(z)[0], keeping what the backward pass needs
    dz = gg.backward(dimg)                              # [n, k, D] = d<img, dimg>/dz

Weights are constants.  Per modulated layer  c_o = d_o * sum_i s_i (W_oi * x_i)  the pass needs (csrc/backward.hip):
  * dx_i = s_i g_i with g_i = sum_o W_oi^T * (d_o dc_o): the forward tap-list MFMA kernel on channel-transposed taps, the
    demodulation coefficients riding on its input-scale port (the stride-2 transposed conv's gradient is a stride-2 conv);
  * ds_i = <x_i, g_i> - s_i sum_o <dc_o, c_o> d_o^2 wsq[o,i]: two per-channel dot products, no per-sample weight-gradient GEMM;
  * FIR gradients = mgf_upfirdn2d with up/down swapped (upfirdn2d.py:237-256); the duplex attention has its own backward kernel.
Activations are the forward's own buffers (nothing is stored twice); the pre-activation of a fused conv+bias+lrelu layer is
recovered by inverting the leaky ReLU.
"""
from __future__ import annotations

import math
import os
from ctypes import sizeof as ctypes_sizeof

import torch

from . import _lib
from . import conv as cv
from .engine import Generator

GRAD_FUSE_SKIP = os.environ.get("MGF_GRAD_FUSE_SKIP", "1") != "0"
FUSE_STYLE_ACT = os.environ.get("MGF_FUSE_STYLE_ACT", "1") != "0"      # tuning / test hook: 0 = style_grad and act_bwd as two launches
# the attention layers' value gradients and demodulation dot products -- by-products nothing on the critical path waits for -- for all layers
# in two launches at the end of the pass (0: three small launches per layer where they arise; tuning / equivalence test)
DEFER_ATTN_GRADS = os.environ.get("MGF_DEFER_ATTN_GRADS", "1") != "0"
# the resnet skip branch's gradient (2x FIR down-sampling + 1x1 conv on transposed weights: memory bound, depends only on the block's output
# gradient) on the generator's side stream next to the block's main chain, from this map size up; it then writes d(x_in) first and the main
# chain's style-gradient kernel accumulates into it.  0 = in line, after the main chain (a forked graph branch costs ~10 us on this runtime:
# it only pays where the branch is long)
SKIP_STREAM_MIN_RES = int(os.environ.get("MGF_GRAD_SKIP_STREAM", "0"))
# up-sampling layers without attention (256^2 and larger): conv1's style gradient, conv0's activation backward AND the adjoint of conv0's blur in
# one pass (mgf_style_act_fir_bwd_f32) -- conv0's dz never reaches memory; 0 = the blur's gradient as its own upfirdn2d launch (equivalence test)
FUSE_ACT_FIR = os.environ.get("MGF_FUSE_ACT_FIR", "1") != "0"


class GeneratorGrad:
    def __init__(self, G: Generator):
        self.G = G
        P = G.plan
        self.T = {}
        for lp in P.layers:
            # 3x3 stride-1 correlation -> true convolution (taps reversed); the transposed conv's gradient is a plain stride-2
            # correlation with the same taps; 1x1 needs no flip
            self.T[lp.name] = cv.transpose_packed(lp.pc, flip=(lp.kind == "conv3"))
        self.Tskip = {res: cv.transpose_packed(pc, flip=False) for res, pc in P.skips.items()}
        # layers the forward runs through the Winograd kernel take it for their data gradient too: the transposed, flipped kernel
        # is rebuilt from the packed taps (gain already folded in) and transformed once
        self.Tw = {}
        for lp in P.layers:
            if lp.wino_u is not None:
                w = lp.pc.wp[:, :, :lp.cout].reshape(3, 3, lp.cin, lp.cout).permute(3, 2, 0, 1)        # [cout, cin, kh, kw]
                self.Tw[lp.name] = cv.winograd_pack(w.permute(1, 0, 2, 3).flip(2, 3).contiguous(), 1.0, lp.res)
        self._bufs = {}
        self._n = None
        self._gen = None                  # generation id of the generator workspace the job tables below point into
        self.per_layer = False
        self.map_scratch = None
        self.z = None
        self.psi = 1.0
        self.debug = None                 # dict -> clones of the intermediate gradients (tests / tools only)

    # ------------------------------------------------------------------ workspace
    def buf(self, role, shape):
        key = (role, tuple(shape))
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(shape, dtype=torch.float32, device=self.G.device)
            self._bufs[key] = t
        return t

    def _alloc(self, n):
        """Per-batch-size reduction buffers and the device job tables of the latent-side kernels."""
        G, P, cfg, L = self.G, self.G.plan, self.G.cfg, _lib.lib()
        D, T = cfg.w_dim, cfg.k - 1
        self._n, self._gen = n, G.ws_gen
        self._fir_mode = self._want_fir_mode()
        # per-slice partial sums of mgf_attn_values_grad_ws (allocated here, not at first use: never inside a captured launch sequence)
        cmax = max([lp.attn.c for lp in P.layers if lp.attn is not None] or [1])
        self.avg_ws = torch.empty(int(L.mgf_attn_values_grad_workspace_floats(n, cmax)), dtype=torch.float32, device=G.device)
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=G.device)
        self.ds_part, self.dc_part, self.dvwb = {}, {}, {}
        sj, aj = [], []
        by_name = {lp.name: lp for lp in P.layers}
        for lp in P.layers:
            in_res = lp.res // lp.up
            sc, dc = int(L.mgf_bwd_chunks(in_res * in_res)), int(L.mgf_bwd_chunks(lp.res * lp.res))
            # the fused style / activation / blur-gradient pass leaves one partial per 64 x 64 tile instead of one per 4096-element chunk:
            # conv1's <x, g> partials and conv0's <dz, c> partials of the blocks it serves
            if self._fir_mode and lp.name.endswith(".conv1") and self._fir_block(by_name.get(lp.name[:-1] + "0"), lp):
                sc = int(L.mgf_style_act_fir_tiles(lp.res, lp.res))
            if self._fir_mode and lp.name.endswith(".conv0") and self._fir_block(lp, by_name.get(lp.name[:-1] + "1")):
                dc = int(L.mgf_style_act_fir_tiles(lp.res, lp.res))
            self.ds_part[lp.name] = e(n, lp.cin, sc)
            if lp.demod:
                self.dc_part[lp.name] = e(n, lp.cout, dc)
            sj.append(_lib.StyleBwdJob(lp.aff_w.data_ptr(), _lib.ptr(lp.pc.wsq) if lp.demod else 0, G._s(lp).data_ptr(),
                                       G._d(lp).data_ptr() if lp.demod else 0, self.ds_part[lp.name].data_ptr(),
                                       self.dc_part[lp.name].data_ptr() if lp.demod else 0, lp.cin, lp.cout, sc, dc,
                                       1.0 / math.sqrt(D), lp.style_gain))
            if lp.attn is not None:
                self.dvwb[lp.name] = e(n, lp.attn.c, T)
                aj.append(_lib.AttnBwdJob(lp.attn.wmv.data_ptr(), self.dvwb[lp.name].data_ptr(), lp.attn.c, 0))
        self._build_deferred_attention(n)
        arr = (_lib.StyleBwdJob * len(sj))(*sj)
        self.style_jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(G.device)
        self.n_style_jobs = len(sj)
        self.attn_jobs, self.n_attn_jobs = None, len(aj)
        if aj:
            arr = (_lib.AttnBwdJob * len(aj))(*aj)
            self.attn_jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(G.device)
        self.dwg = e(n, len(sj), D)
        self.dyc = e(n, max(len(aj), 1), T, D)
        self.dw = e(n, cfg.k, D)
        self.dz = e(n, cfg.k, D)
        # W+ (per-layer latents, networks.py:1252-1253): layer j's style gradient belongs to slot lp.slot of the global component, its
        # attention-value gradient to the same slot of the T local components -- index tables for the scatter of backward_ws
        self.dws = e(n, cfg.k, cfg.num_ws, D)
        self.style_slots = torch.tensor([lp.slot for lp in P.layers], dtype=torch.int64, device=G.device)
        self.attn_slots = torch.tensor([lp.slot for lp in P.layers if lp.attn is not None], dtype=torch.int64, device=G.device)
        self.max_channels = max(max(lp.cin, lp.cout) for lp in P.layers)

    def _want_fir_mode(self):
        return bool(FUSE_ACT_FIR and FUSE_STYLE_ACT and self.debug is None)

    @staticmethod
    def _fir_block(l0, l1):
        """Does the (conv0, conv1) pair of a block take the fused style / activation / blur-gradient pass?  conv0 an up-sampling layer with
        bias / activation and no attention, 16-byte rows."""
        return (l0 is not None and l1 is not None and l0.kind == "tconv" and l0.attn is None and l1.attn is None and l0.bias is not None
                and l0.res % 4 == 0 and l0.res >= 64)

    def _build_deferred_attention(self, n):
        """Per-layer buffers (dc, dg, probabilities, slice partials) and the device job table of mgf_attn_grad_multi.  The buffers must
        outlive their layer's step of the pass -- 2 x the attention layers' activations (110 MB per sample at 1024^2) instead of one shared
        set -- which buys 33 small launches for 2."""
        G, P, L = self.G, self.G.plan, _lib.lib()
        T = G.cfg.k - 1
        self.attn_defer = None
        layers = [lp for lp in P.layers if lp.attn is not None]
        if not (DEFER_ATTN_GRADS and layers):
            return
        assert ctypes_sizeof(_lib.AttnGradJob) == int(L.mgf_attn_grad_job_bytes())
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=G.device)
        jobs, bufs, g0, d0, r0 = [], {}, 0, 0, 0
        for lp in layers:
            a = lp.attn
            side = lp.res
            slices = int(L.mgf_attn_values_grad_slices(n, a.c, a.f, T))
            if slices <= 0:
                return                                  # a shape the MFMA form does not take: keep the per-layer launches for all
            nchunk = int(L.mgf_bwd_chunks(a.f))
            b = dict(dc=e(n, a.c, side, side), dg=e(n, a.c, side, side), probs=e(n, a.f, T), part=e(n * slices * a.c * 16))
            bufs[lp.name] = b
            jobs.append(_lib.AttnGradJob(b["dg"].data_ptr(), b["probs"].data_ptr(), b["dc"].data_ptr(), 0, b["part"].data_ptr(),
                                         self.dvwb[lp.name].data_ptr(), self.dc_part[lp.name].data_ptr() if lp.demod else 0, a.c, a.f, slices,
                                         nchunk, g0, d0, r0, 0))
            g0 += slices * -(-a.c // 128) * n
            d0 += (nchunk * a.c * n) if lp.demod else 0
            r0 += -(-(a.c * 16) // 256) * n
        self.attn_defer = dict(jobs=jobs, bufs=bufs, blocks=(g0, d0, r0), table=None, cpre={})

    def _deferred_attention_launch(self):
        """The two launches of the deferred attention by-products (after the last layer of the pass)."""
        D = self.attn_defer
        if D["table"] is None or D["cpre_ptrs"] != tuple(D["cpre"][j.dg] for j in D["jobs"]):
            # (the conv outputs c_pre live in the generator's workspace: their addresses are known once the pass has walked the layers)
            for j in D["jobs"]:
                j.cpre = D["cpre"][j.dg]
            D["cpre_ptrs"] = tuple(j.cpre for j in D["jobs"])
            arr = (_lib.AttnGradJob * len(D["jobs"]))(*D["jobs"])
            D["table"] = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.G.device)
        g, d, r = D["blocks"]
        _lib.check(_lib.lib().mgf_attn_grad_multi(D["table"].data_ptr(), len(D["jobs"]), self.G.n, g, d, r, _lib.stream_ptr()), "attn_grad_multi")

    # ------------------------------------------------------------------ forward
    def forward(self, z=None, ws=None, truncation_psi=1, noise_mode="const", noises=None):
        """G(z)[0] (or G(ws=...)) with conv_last kept in memory (the fused conv_last+ToRGB kernel never writes it)."""
        G = self.G
        if ws is not None:
            # ws [n, k, num_ws, D]: a broadcast VIEW (what mapping() returns: stride 0 over the layer axis) or [n, k, D] takes the shared
            # tables; anything dense takes the per-layer tables -- always correct, also for a materialised broadcast (G.mapping(z, psi) /
            # return_ws=True hand out dense copies): backward_w() then returns the SUM over the slots, i.e. the gradient with respect to
            # one shared latent set, and backward_ws() the per-slot (W+) gradient
            if ws.ndim == 4 and ws.stride(2) == 0:
                ws = ws[:, :, 0]
            w = ws.contiguous().float()
            self.per_layer = w.ndim == 4
        else:
            self.per_layer = False
        # the backward pass reads conv_last's activation and the full-resolution skip tensors: both fusions off for this forward
        # (the skip branch stays fused into conv1's Winograd epilogue where the engine does that -- G.skip_fused[res] -- and the backward
        # up-samples the half-resolution skip tensor itself; MGF_GRAD_FUSE_SKIP=0: the full-resolution tensor in both passes)
        G.fuse_torgb, keep_up = False, G.fuse_skip_up
        if not GRAD_FUSE_SKIP:
            G.fuse_skip_up = False
        if z is not None:
            # the mapping network's activations are kept for backward() (mgf_mapping_backward_saved) rather than recomputed there
            need = z.shape[0] * int(_lib.lib().mgf_mapping_bwd_scratch_floats(G.cfg.k, G.cfg.w_dim, G.cfg.mapping_layers // 2))
            if self.map_scratch is None or self.map_scratch.numel() != need:
                self.map_scratch = torch.empty(need, device=G.device, dtype=torch.float32)
            G.map_save = self.map_scratch
        try:
            if ws is not None:
                if (w.shape[0], False) != (G.n, G.lean):        # the backward pass reads every layer output: the FULL workspace flavour
                    G._alloc(w.shape[0], False)
                img = G.synthesis(w, noise_mode=noise_mode, noises=noises)
            else:
                img = G.forward_workspace(z, None, truncation_psi=truncation_psi, noise_mode=noise_mode, noises=noises)[0]
        finally:
            G.fuse_torgb, G.fuse_skip_up, G.map_save = True, keep_up, None
        self.z = None if z is None else z.contiguous().float()
        self.psi = float(truncation_psi) if ws is None else 1.0
        if (self._n, self._gen, getattr(self, "_fir_mode", None)) != (G.n, G.ws_gen, self._want_fir_mode()):
            # after G: its style / demod arenas (pointed to by the job tables) are sized by then; a workspace evicted and re-created at the same
            # batch size has a new generation id; the partial-sum layout depends on which fused passes run
            self._alloc(G.n)
        return img

    # ------------------------------------------------------------------ backward
    def _act_bwd(self, lp, dy, y_out, residual, residual_low=None):
        """d(pre-activation) of one SynthesisLayer from the gradient dy of its output y_out = lrelu(c [attention] + noise + bias) * gain
        + residual (and, for a demodulated layer without attention, the <dz, c> partials of the demodulation gradient)."""
        G, L, st = self.G, _lib.lib(), _lib.stream_ptr()
        n, c, h, w = y_out.shape
        mode, noises = G.last_noise
        noise, noise_n = G._noise_for(lp, mode, noises)
        dz = self.buf("dz", y_out.shape)
        want_dot = lp.demod and lp.attn is None
        # conv_last carries neither noise nor bias/activation (engine.synthesis): alpha = gain = 1 makes the kernel a plain copy + dot
        alpha, gain = (0.2, lp.act_gain) if lp.bias is not None else (1.0, 1.0)
        # residual_low: the skip branch at HALF resolution, as the forward's Winograd epilogue consumed it (engine fuse_skip_up): it is
        # up-sampled inside the kernel and the full-resolution skip tensor exists in neither pass
        _lib.check(L.mgf_layer_act_bwd_low_f32(dz.data_ptr(), self.dc_part[lp.name].data_ptr() if want_dot else None, dy.data_ptr(),
                                               y_out.data_ptr(), _lib.ptr(residual), _lib.ptr(residual_low), w, _lib.ptr(lp.bias),
                                               _lib.ptr(noise), _lib.ptr(lp.noise_strength) if noise is not None else None, noise_n, n, c,
                                               h * w, alpha, gain, st), "layer_act_bwd")
        return dz

    def _conv_bwd(self, lp, dz, y_out, c_pre, x_in, dT=None):
        """dz -> g = the data gradient of the layer's convolution BEFORE the style scale (d(x_in) = s * g): attention backward when the
        layer has one, then the convolution on the transposed taps.  dT: the blur's adjoint of dz, already computed by the fused pass
        (_style_act_fir_bwd; dz is then None)."""
        G, L, st = self.G, _lib.lib(), _lib.stream_ptr()
        n, c, h, w = y_out.shape
        hw = h * w
        if dT is not None:
            return cv.conv_forward(dT, self.T[lp.name], stride=2, pad=(0, 0), in_scale=G._d(lp) if lp.demod else None, out=self.buf("g", x_in.shape))
        dc = dz
        if lp.attn is not None and self.attn_defer is not None and self.debug is None:
            # deferred form: this layer's dc / dg / probabilities stay in buffers of their own; the value gradient and the demodulation
            # dot products of ALL attention layers are two launches at the end of the pass (_deferred_attention_launch)
            a = lp.attn
            T = G.cfg.k - 1
            b = self.attn_defer["bufs"][lp.name]
            dc = b["dc"]
            _lib.check(L.mgf_duplex_attention_bwd(dc.data_ptr(), b["dg"].data_ptr(), b["probs"].data_ptr(), dz.data_ptr(), c_pre.data_ptr(),
                                                  a.wqc.data_ptr(), a.spos.data_ptr(), G._v(lp).data_ptr(), n, a.c, a.f, T, st),
                       "duplex_attention_bwd")
            self.attn_defer["cpre"][b["dg"].data_ptr()] = c_pre.data_ptr()
        elif lp.attn is not None:
            a = lp.attn
            T = G.cfg.k - 1
            dc, dg, probs = self.buf("dc", y_out.shape), self.buf("dg", y_out.shape), self.buf("probs", (n, a.f, T))
            _lib.check(L.mgf_duplex_attention_bwd(dc.data_ptr(), dg.data_ptr(), probs.data_ptr(), dz.data_ptr(), c_pre.data_ptr(),
                                                  a.wqc.data_ptr(), a.spos.data_ptr(), G._v(lp).data_ptr(), n, a.c, a.f, T, st),
                       "duplex_attention_bwd")
            ws = self.avg_ws
            _lib.check(L.mgf_attn_values_grad_ws(self.dvwb[lp.name].data_ptr(), dg.data_ptr(), probs.data_ptr(), n, a.c, a.f, T,
                                                 ws.data_ptr(), ws.numel(), st), "attn_values_grad")
            if lp.demod:
                _lib.check(L.mgf_channel_dot_f32(self.dc_part[lp.name].data_ptr(), dc.data_ptr(), c_pre.data_ptr(), n, c, hw, st),
                           "channel_dot")
        if self.debug is not None:
            self.debug[lp.name + ":dc"] = dc.clone()
        d = G._d(lp) if lp.demod else None
        if lp.kind == "tconv":
            dT = self.buf("dT", (n, c, h + 1, w + 1))
            cv.upfirdn_into(dT, dc, G.plan.fir, up=1, pad=(2, 2, 2, 2), gain=4.0, flip=True, separable=True)    # (plan.fir is an outer product, cf. engine._layer)
            return cv.conv_forward(dT, self.T[lp.name], stride=2, pad=(0, 0), in_scale=d, out=self.buf("g", x_in.shape))
        pad = (1, 1) if lp.kind == "conv3" else (0, 0)
        if lp.name in self.Tw and cv.winograd_fills_chip(n, x_in.shape[1], h, w):
            return cv.winograd_forward(dc, self.Tw[lp.name], in_scale=d, out=self.buf("g", x_in.shape))
        return cv.conv_forward(dc, self.T[lp.name], pad=pad, in_scale=d, out=self.buf("g", x_in.shape))

    def _style_bwd(self, lp, g, x_in, dx_role, accumulate=False):
        """<x_in, g> partials of the style gradient and d(x_in) (+)= s * g."""
        G, L, st = self.G, _lib.lib(), _lib.stream_ptr()
        n = x_in.shape[0]
        dx = self.buf(dx_role, x_in.shape)
        ci, hi = x_in.shape[1], x_in.shape[2] * x_in.shape[3]
        _lib.check(L.mgf_style_grad_f32(self.ds_part[lp.name].data_ptr(), dx.data_ptr(), x_in.data_ptr(), g.data_ptr(),
                                        G._s(lp).data_ptr(), n, ci, hi, int(accumulate), st), "style_grad")
        return dx

    def _style_act_bwd(self, lp, g, prev, y_prev, residual=None, dx_role=None, residual_low=None):
        """_style_bwd of `lp` fused with _act_bwd of the layer `prev` whose output y_prev is lp's input: one pass over (y_prev, g)
        (mgf_style_grad_act_bwd_f32).  conv1 -> conv0 of a block: prev has no residual and the intermediate s * g never reaches memory.
        conv_last -> conv1 of the last block: prev's residual is the block's skip tensor and s * g is stored too (dx_role: the skip
        branch's backward reads it).  Returns dz, or (dz, dx) with a residual."""
        G, L, st = self.G, _lib.lib(), _lib.stream_ptr()
        n, c, h, w = y_prev.shape
        mode, noises = G.last_noise
        noise, noise_n = G._noise_for(prev, mode, noises)
        dz = self.buf("dz", y_prev.shape)
        want_dot = prev.demod and prev.attn is None
        alpha, gain = (0.2, prev.act_gain) if prev.bias is not None else (1.0, 1.0)
        with_res = residual is not None or residual_low is not None
        dx = self.buf(dx_role, y_prev.shape) if with_res else None
        _lib.check(L.mgf_style_grad_act_bwd_f32(self.ds_part[lp.name].data_ptr(), self.dc_part[prev.name].data_ptr() if want_dot else None,
                                                dz.data_ptr(), _lib.ptr(dx), y_prev.data_ptr(), g.data_ptr(), G._s(lp).data_ptr(),
                                                _lib.ptr(residual), _lib.ptr(residual_low), w, _lib.ptr(prev.bias), _lib.ptr(noise),
                                                _lib.ptr(prev.noise_strength) if noise is not None else None, noise_n,
                                                n, c, h * w, alpha, gain, st), "style_grad_act_bwd")
        return (dz, dx) if with_res else dz

    def _style_act_fir_bwd(self, lp, g, prev, y_prev):
        """_style_act_bwd of (lp, prev) followed by the adjoint of prev's blur, in one pass (mgf_style_act_fir_bwd_f32): returns
        dT [n, c, h + 1, w + 1], the map prev's stride-2 data-gradient convolution reads."""
        G, L, st = self.G, _lib.lib(), _lib.stream_ptr()
        n, c, h, w = y_prev.shape
        mode, noises = G.last_noise
        noise, noise_n = G._noise_for(prev, mode, noises)
        dT = self.buf("dT", (n, c, h + 1, w + 1))
        _lib.check(L.mgf_style_act_fir_bwd_f32(self.ds_part[lp.name].data_ptr(), self.dc_part[prev.name].data_ptr() if prev.demod else None,
                                               dT.data_ptr(), y_prev.data_ptr(), g.data_ptr(), G._s(lp).data_ptr(), _lib.ptr(prev.bias),
                                               _lib.ptr(noise), _lib.ptr(prev.noise_strength) if noise is not None else None, noise_n,
                                               G.plan.fir.data_ptr(), 1, 4.0, n, c, h, w, 0.2, prev.act_gain, st), "style_act_fir_bwd")
        return dT

    def _layer_bwd(self, lp, dy, y_out, c_pre, residual, x_in, dx_role):
        """One SynthesisLayer.  dy: gradient wrt the layer output y_out (= lrelu(c [attention] + noise + bias) * gain + residual);
        c_pre: the stored demodulated conv output when the layer has attention; x_in: the layer input.  Returns d(x_in)."""
        dz = self._act_bwd(lp, dy, y_out, residual)
        g = self._conv_bwd(lp, dz, y_out, c_pre, x_in)
        return self._style_bwd(lp, g, x_in, dx_role)

    def backward_w(self, dimg):
        """dimg [n,3,R,R] -> dw [n,k,D]: gradient of <img, dimg> with respect to the intermediate latent w (after a forward with
        per-layer ws: summed over the layer slots, i.e. with respect to a latent set shared by all layers)."""
        G, cfg, L = self.G, self.G.cfg, _lib.lib()
        self._backward_layers(dimg)
        _lib.check(L.mgf_latent_grad_gather(self.dw.data_ptr(), self.dwg.data_ptr(), self.n_style_jobs, self.dyc.data_ptr(),
                                            self.n_attn_jobs, G.n, cfg.k, cfg.w_dim, self.psi, _lib.stream_ptr()), "latent_grad_gather")
        return self.dw

    def backward_ws(self, dimg):
        """dimg [n,3,R,R] -> dws [n,k,num_ws,D]: the W+ gradient -- every layer's latent gradient lands in its own slot (layer `slot`
        reads ws[:, :, slot] for its style AND its attention values, networks.py:1022-1031,1252-1253; slots are unique per layer, so the
        scatter has no collisions).  Valid after forward(ws=...) or forward(z): summed over axis 2 it equals backward_w()."""
        G, cfg = self.G, self.G.cfg
        self._backward_layers(dimg)
        T = cfg.k - 1
        self.dws.zero_()
        self.dws[:, T].index_copy_(1, self.style_slots, self.dwg)                                  # [n, num_ws, D] <- [n, layers, D]
        if self.n_attn_jobs:
            self.dws[:, :T].index_copy_(2, self.attn_slots, self.dyc[:, :self.n_attn_jobs].permute(0, 2, 1, 3))     # [n, T, num_ws, D] <- [n, T, A, D]
        if self.psi != 1.0:
            self.dws.mul_(self.psi)
        return self.dws

    def _backward_layers(self, dimg):
        """The synthesis network's backward pass down to the per-layer latent gradients: dwg [n, layers, D] (style path -> the global
        component) and dyc [n, attention layers, T, D] (attention values -> the local components)."""
        G, P, cfg, L = self.G, self.G.plan, self.G.cfg, _lib.lib()
        _lib.require_gpu(dimg)
        n = G.n
        assert self._n == n, "call forward() first"
        assert tuple(dimg.shape) == tuple(G.img.shape) and dimg.dtype == torch.float32
        dimg = dimg.contiguous()
        st = _lib.stream_ptr()
        D, T = cfg.w_dim, cfg.k - 1
        layers = {lp.name: lp for lp in P.layers}
        R = cfg.img_resolution
        out_of = lambda res: G.bufs[res]["conv1a" if cfg.has_attention(res) else "conv1"]
        b = f"synthesis.b{R}"
        lt, ll = layers[b + ".torgb"], layers[b + ".conv_last"]
        h = G.bufs[R]["last"]
        dz1_top = None
        # ToRGB: img = sum_co W[c,co] s[co] h[co] + bias (no demodulation, no activation)
        g = cv.conv_forward(dimg, self.T[lt.name], out=self.buf("g", h.shape))
        if FUSE_STYLE_ACT and self.debug is None:
            # ToRGB's input IS conv_last's output h: its style gradient and conv_last's (linear) activation backward in one pass over (h, g)
            dzl = self._style_act_bwd(lt, g, ll, h)
            gl = self._conv_bwd(ll, dzl, h, None, out_of(R))
            if R > 4:
                # conv_last's input IS the last block's output (conv1's activation + the skip tensor): conv_last's style gradient and
                # conv1's activation backward in one pass as well; s g is still stored, the skip branch's backward reads it
                low = G.skip_fused.get(R, False)
                dz1_top, dx = self._style_act_bwd(ll, gl, layers[b + ".conv1"], out_of(R), dx_role="dxin",
                                                  residual=None if low else G.bufs[R]["skip"], residual_low=G.bufs[R]["skip_low"] if low else None)
            else:
                dx = self._style_bwd(ll, gl, out_of(R), "dxin")
        else:
            dh = self.buf("dh", h.shape)
            _lib.check(L.mgf_style_grad_f32(self.ds_part[lt.name].data_ptr(), dh.data_ptr(), h.data_ptr(), g.data_ptr(),
                                            G._s(lt).data_ptr(), n, h.shape[1], h.shape[2] * h.shape[3], 0, st), "style_grad(torgb)")
            dx = self._layer_bwd(ll, dh, h, None, None, out_of(R), "dxin")
        for res in reversed(cfg.block_resolutions):
            B = G.bufs[res]
            att = cfg.has_attention(res)
            l1 = layers[f"synthesis.b{res}.conv1"]
            y1 = B["conv1a"] if att else B["conv1"]
            if res == 4:
                self._layer_bwd(l1, dx, y1, B["conv1"] if att else None, None, G.const_in, "dxin")
                break
            l0 = layers[f"synthesis.b{res}.conv0"]
            y0 = B["conv0a"] if att else B["conv0"]
            x_prev = out_of(res // 2)
            d_out = dx
            if self.debug is not None:
                self.debug[f"synthesis.b{res}:dout"] = d_out.clone()
            side = bool(SKIP_STREAM_MIN_RES) and res >= SKIP_STREAM_MIN_RES and self.debug is None
            if side:
                # skip branch first, on the side stream: y = upfirdn(conv1x1(x), up=2)  ->  d(x) = conv1x1^T(upfirdn(dy, down=2)) written
                # into d(x_in); the main chain's style-gradient kernel adds its s g to it after the join
                main = torch.cuda.current_stream(G.device)
                G.side.wait_stream(main)
                with torch.cuda.stream(G.side):
                    dlow = self.buf("dlow", B["skip_low"].shape)
                    cv.upfirdn_into(dlow, d_out, P.fir, up=1, down=2, pad=(1, 1, 1, 1), gain=4.0, flip=True)
                    cv.conv_forward(dlow, self.Tskip[res], out=self.buf("dxin", x_prev.shape))
            # conv1 then conv0: conv1's input IS conv0's output y0, so conv1's style gradient and conv0's activation backward are one
            # pass over (y0, g) -- d(y0) = s g is never stored (MGF_FUSE_STYLE_ACT=0: the two kernels in sequence, bit-identical)
            if res == R and dz1_top is not None:
                dz1 = dz1_top
            elif G.skip_fused.get(res, False):
                dz1 = self._act_bwd(l1, d_out, y1, None, residual_low=B["skip_low"])
            else:
                dz1 = self._act_bwd(l1, d_out, y1, B["skip"])
            g1 = self._conv_bwd(l1, dz1, y1, B["conv1"] if att else None, y0)
            dT0 = None
            if self._fir_mode and self._fir_block(l0, l1):
                dT0 = self._style_act_fir_bwd(l1, g1, l0, y0)       # conv0's dz goes straight through the blur's adjoint: never stored
                dz0 = None
            elif FUSE_STYLE_ACT and self.debug is None:
                dz0 = self._style_act_bwd(l1, g1, l0, y0)
            else:
                dmid = self._style_bwd(l1, g1, y0, "dmid")
                if self.debug is not None:
                    self.debug[f"synthesis.b{res}:dmid"] = dmid.clone()
                dz0 = self._act_bwd(l0, dmid, y0, None)
            g0 = self._conv_bwd(l0, dz0, y0, B["conv0"] if att else None, x_prev, dT=dT0)
            if side:
                main.wait_stream(G.side)
                dxin = self._style_bwd(l0, g0, x_prev, "dxin", accumulate=True)
            else:
                dxin = self._style_bwd(l0, g0, x_prev, "dxin")
                # skip branch: y = upfirdn(conv1x1(x), up=2, pad (2,1,2,1), gain 4)  ->  conv1x1^T(upfirdn(dy, down=2, pad (1,1,1,1)))
                dlow = self.buf("dlow", B["skip_low"].shape)
                cv.upfirdn_into(dlow, d_out, P.fir, up=1, down=2, pad=(1, 1, 1, 1), gain=4.0, flip=True)
                cv.conv_forward(dlow, self.Tskip[res], epilogue=_lib.make_epilogue(residual=dxin), out=dxin)
            dx = dxin
        if self.attn_defer is not None and self.debug is None:
            self._deferred_attention_launch()
        # d(styles) -> the global component and d(attention values) -> the local components: one launch (two independent latency chains)
        _lib.check(L.mgf_latent_bwd_multi(self.dwg.data_ptr(), self.style_jobs.data_ptr(), self.n_style_jobs, self.dyc.data_ptr(),
                                          _lib.ptr(self.attn_jobs), self.n_attn_jobs, n, T, D, self.max_channels, st), "latent_bwd_multi")

    def backward(self, dimg):
        """dimg -> dz [n,k,D] (through the mapping network; forward() must have been called with z)."""
        G, cfg, L = self.G, self.G.cfg, _lib.lib()
        assert self.z is not None, "forward() was called with ws=...: use backward_w"
        dw = self.backward_w(dimg)
        _lib.check(L.mgf_mapping_backward_saved(self.dz.data_ptr(), dw.data_ptr(), self.z.data_ptr(), G.plan.mapping_blob.data_ptr(),
                                          self.map_scratch.data_ptr(), G.n, cfg.k, cfg.w_dim, cfg.mapping_layers // 2,
                                          int(cfg.normalize_global), _lib.stream_ptr()), "mapping_backward_saved")
        return self.dz
