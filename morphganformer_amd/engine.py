"""GANformer generator forward on MI355X: checkpoint -> device plan -> kernel schedule.

Replaces Generator.forward / MappingNetwork / SynthesisNetwork (training/networks.py:894-942, 1244-1264, 1304-1331) for the
configuration the projection drivers use (resnet architecture, duplex attention with parametric centroids, integration
"mul", layer norm).  All arithmetic is float32 on the device; everything that depends only on the
checkpoint is folded once, in float64, at plan-build time:

  * conv weights -> tap-major images (+ the sum-of-squares table that turns demodulation into a small GEMV),
  * attention: query projection x positional term x att_weight x centroids x 1/sqrt(C) -> wqc [C,T] and spos [F,T];
    value projection x modulation FC (+ bias + 1) -> wmv [C,D], bmv [C]          (no F x C x C GEMM is left),
  * mapping network: learning-rate multipliers, He gains, positional terms and the score scale folded into one blob.

Per call the schedule is: mapping (1 launch) -> styles/demod for all 20 modulated layers (1 launch) -> attention value
tables for all attention layers (1 launch) -> per block {skip 1x1, skip FIR-up, transposed conv, FIR (+epilogue),
[attention], conv, [attention]} -> conv_last -> toRGB.  No host synchronisation, no allocation: graph-capturable.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from . import conv as cv
from .synth_weights import GeneratorConfig

SQRT2 = math.sqrt(2.0)
SQRT_HALF = math.sqrt(0.5)
USE_WINOGRAD = os.environ.get("MGF_WINOGRAD", "1") != "0"
WINOGRAD_MAX_RES = int(os.environ.get("MGF_WINOGRAD_MAX_RES", "1024"))
BF_DIRECT_MAX_RES = int(os.environ.get("MGF_BF_DIRECT_MAX_RES", "128"))     # arith="bf16x3": largest map whose 3x3 layers leave Winograd (tuning hook)


def pack_mapping_params(sd, cfg: GeneratorConfig) -> np.ndarray:
    """Flatten the mapping network into the blob read by mgf_mapping_forward (csrc/latent_prep.hip: mapping_kernel).

    Layout (float32, matrices [out][in] row-major, all gains folded):
      global mlp : n_res x {W0, b0, W1, b1}, Wout, bout
      local  mlp : n_res x {Wq*, bq_pos[T,D]*, Wk, bk_pos[T,D], Wv, bv, Wm, bm, W0, b0, W1, b1}, Wout, bout
    (*) pre-multiplied by the 1/sqrt(D) score scale; b?_pos = bias + positional projection of G.pos.
    """
    D, T = cfg.w_dim, cfg.k - 1
    lr = cfg.mapping_lrmul
    n_res = cfg.mapping_layers // 2
    g64 = lambda k: np.asarray(sd[k], dtype=np.float64)
    out = []

    def fc(prefix, lrmul):
        w = g64(prefix + ".weight")
        return w * (lrmul / math.sqrt(w.shape[1])), g64(prefix + ".bias") * lrmul

    for i in range(n_res):
        for j in (0, 1):
            w, b = fc(f"mapping.global_mlp.l{i}.fc{j}", lr)
            out += [w, b]
    out += list(fc("mapping.global_mlp.out_layer", lr))
    pos = g64("pos")
    for i in range(n_res):
        p = f"mapping.mlp.sa{i}"
        wq, bq = fc(p + ".to_queries", 1.0)
        wfp, bfp = fc(p + ".from_pos_map", 1.0)
        wk, bk = fc(p + ".to_keys", 1.0)
        wtp, btp = fc(p + ".to_pos_map", 1.0)
        wv, bv = fc(p + ".to_values", 1.0)
        wm, bm = fc(p + ".modulation", 1.0)
        sc = 1.0 / math.sqrt(D)
        out += [wq * sc, (bq[None] + pos @ wfp.T + bfp[None]) * sc, wk, bk[None] + pos @ wtp.T + btp[None], wv, bv, wm, bm]
        for j in (0, 1):
            w, b = fc(f"mapping.mlp.l{i}.fc{j}", lr)
            out += [w, b]
    out += list(fc("mapping.mlp.out_layer", lr))
    blob = np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1) for a in out]).astype(np.float32)
    expect = _lib.lib().mgf_mapping_param_floats(cfg.k, D, n_res)
    assert blob.size == expect, (blob.size, expect)
    return blob


@dataclass
class ConvLayerPlan:
    name: str
    res: int
    cin: int
    cout: int
    up: int
    slot: int
    kind: str                       # "conv3" | "tconv" | "torgb"
    pc: cv.PackedConv = None
    aff_w: torch.Tensor = None
    aff_b: torch.Tensor = None
    style_gain: float = 1.0
    demod: bool = True
    bias: torch.Tensor = None
    noise_const: torch.Tensor = None
    noise_strength: torch.Tensor = None
    act_gain: float = 1.0
    attn: "AttnPlan" = None
    w_raw: torch.Tensor = None      # toRGB only: [img_channels, cin] un-packed weights for the fused projection
    wino_u: torch.Tensor = None     # 3x3 stride-1 layers on 16^2 .. 256^2 maps: Winograd-transformed weights (csrc/wino.hip)
    pcb: torch.Tensor = None        # arith="bf16x3" only: the weights split into two bfloat16 terms (conv.pack_weights_bf16x3)
    s_off: int = 0                  # offsets (floats) into the per-sample style / demod arenas
    d_off: int = 0


@dataclass
class AttnPlan:
    c: int
    f: int
    wqc: torch.Tensor               # [C, T]
    spos: torch.Tensor              # [F, T]
    wmv: torch.Tensor               # [C, D]
    bmv: torch.Tensor               # [C]
    v_off: int = 0                  # offset into the per-sample value-table arena


class SynthesisPlan:
    """Device-resident, checkpoint-derived constants of one generator."""

    def __init__(self, sd, cfg: GeneratorConfig, device="cuda", arith="f32"):
        _lib.lib()
        self.cfg = cfg
        self.arith = arith
        self.device = torch.device(device)
        dev = self.device
        f64 = lambda k: np.asarray(sd[k], dtype=np.float64)
        t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
        D, T = cfg.w_dim, cfg.k - 1
        self.mapping_blob = t32(pack_mapping_params(sd, cfg))
        self.const = t32(f64("synthesis.b4.const"))
        self.fir = t32(f64(f"synthesis.b{cfg.block_resolutions[-1]}.resample_kernel"))
        # the resample filter every checkpoint of the reference has (networks.py:1113, resample_kernel = [1,3,3,1]): the form-3 Winograd
        # epilogue up-samples the resnet skip branch with it in place (hard-wired taps), any other filter keeps the separate FIR pass
        self.fir_is_1331 = bool(np.allclose(f64(f"synthesis.b{cfg.block_resolutions[-1]}.resample_kernel"),
                                            np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0, rtol=0, atol=1e-7))
        self.layers: list[ConvLayerPlan] = []
        self.skips = {}
        s_off = d_off = v_off = 0
        for (res, name, cin, cout, up, slot, att, has_nb) in cfg.layer_table():
            p = f"synthesis.b{res}.{name}"
            w = f64(p + ".weight")
            kind = "torgb" if name == "torgb" else ("tconv" if up == 2 else "conv3")
            lp = ConvLayerPlan(name=p, res=res, cin=cin, cout=cout, up=up, slot=slot, kind=kind)
            if kind == "torgb":
                # styles * w_gain instead of weight * w_gain, no demodulation (networks.py:1056-1063)
                lp.pc = cv.pack_weights(t32(w), gain=1.0, flip=False)
                lp.w_raw = t32(w.reshape(cout, cin))
                lp.style_gain = 1.0 / math.sqrt(cin)
                lp.demod = False
            else:
                wg = 1.0 / math.sqrt(cin * 9)
                # up=1: correlation (flip_weight=True); up=2: conv_transpose2d on the un-flipped weights
                lp.pc = cv.pack_weights(t32(w), gain=wg, flip=False, want_wsq=True)
                # Winograd F(2x2,3x3) where it beats the 9-tap kernel IN the iteration (rocprofv3 trace, conv1 layers with their residual
                # epilogue): every 3x3 stride-1 layer from 32^2 up (1024^2: +1 % iterations/s, the smallest margin), conv_last with the
                # ToRGB projection fused into its epilogue like the tap-list launch.  MGF_WINOGRAD=0 (tuning hook) = direct kernel
                # everywhere, MGF_WINOGRAD_MAX_RES limits the map size
                if kind == "conv3" and USE_WINOGRAD and cv.winograd_ok(cin, cout, res, res) and res <= WINOGRAD_MAX_RES:
                    lp.wino_u = cv.winograd_pack(t32(w), wg, res)
                # the opt-in bf16x3 arithmetic: every transposed conv and 3x3 layer the kernel serves (maps at least 32 wide, cin % 16 == 0)
                if arith == "bf16x3" and cin % 16 == 0 and res // up >= 32:
                    lp.pcb = cv.pack_weights_bf16x3(lp.pc)
            lp.aff_w = t32(f64(p + ".affine.weight"))
            lp.aff_b = t32(f64(p + ".affine.bias"))
            if (p + ".biasAct.bias") in sd:
                lp.bias = t32(f64(p + ".biasAct.bias"))
            if (p + ".noise_strength") in sd:
                lp.noise_strength = t32(f64(p + ".noise_strength").reshape(1))
                lp.noise_const = t32(f64(p + ".noise_const"))
            lp.act_gain = SQRT2 * (SQRT_HALF if (name == "conv1" and res > 4) else 1.0)
            if att and (p + ".transformer.to_queries.weight") in sd:
                lp.attn = self._fold_attention(sd, p, cout, res, T, D, t32, f64)
                lp.attn.v_off = v_off
                v_off += cout * T
            lp.s_off, lp.d_off = s_off, d_off
            s_off += cin
            d_off += cout
            self.layers.append(lp)
        self.s_total, self.d_total, self.v_total = s_off, d_off, v_off
        for res in cfg.block_resolutions[1:]:
            w = f64(f"synthesis.b{res}.skip.weight")
            # 1x1, no modulation; w_gain and the resnet gain sqrt(1/2) (networks.py:1121-1122) folded into the weights
            self.skips[res] = cv.pack_weights(t32(w), gain=SQRT_HALF / math.sqrt(w.shape[1]), flip=False)
        torch.cuda.synchronize(dev)

    @staticmethod
    def _fold_attention(sd, p, C_, res, T, D, t32, f64):
        tp = p + ".transformer"
        Fn = res * res
        wq = f64(tp + ".to_queries.weight") / math.sqrt(C_)
        bq = f64(tp + ".to_queries.bias")
        wfp = f64(tp + ".from_pos_map.weight") / math.sqrt(D)
        bfp = f64(tp + ".from_pos_map.bias")
        wv = f64(tp + ".to_values.weight") / math.sqrt(D)
        bv = f64(tp + ".to_values.bias")
        wm = f64(tp + ".modulation.weight") / math.sqrt(C_)
        bm = f64(tp + ".modulation.bias")
        aw = f64(tp + ".att_weight").reshape(2 * C_)
        cent = f64(tp + ".centroids").reshape(T, 2 * C_)
        a1 = cent[:, :C_] * aw[None, :C_]                 # [T, C]
        a2 = cent[:, C_:] * aw[None, C_:]
        inv = 1.0 / math.sqrt(C_)
        wqc = (wq.T @ a1.T) * inv                          # [C, T]
        pos_term = f64(p + ".grid_pos").reshape(Fn, D) @ wfp.T + bfp[None]     # [F, C]
        spos = (pos_term @ a2.T + (bq @ a1.T)[None]) * inv                      # [F, T]
        wmv = wm @ wv                                      # [C, D]
        bmv = wm @ bv + bm + 1.0
        return AttnPlan(c=C_, f=Fn, wqc=t32(wqc), spos=t32(spos), wmv=t32(wmv), bmv=t32(bmv))


class Generator:
    """Callable mirror of the reference Generator (training/networks.py:1269-1331) backed by the HIP kernels.

    __call__(z=None, c=None, ws=None, truncation_psi=1, ..., noise_mode="random") -> tuple, like the reference.
    """

    def __init__(self, sd, cfg: GeneratorConfig, device="cuda", max_batch: int = 1, arith: str = "f32"):
        """arith: "f32" -- the reference's arithmetic, exact float32 on the FP32 matrix cores (what every parity claim and the headline rest
        on) -- or "bf16x3": an OPT-IN mode in which the transposed convolutions and the 3x3 layers below 256^2 split every float32 operand into
        two bfloat16 terms and run three bf16 matrix instructions per product with float32 accumulation (mgf_conv_taps_bf16x3_f32; per-layer
        error 4 - 5e-6 of the output's range instead of 1e-6, ~2x the matrix rate).  Everything else is unchanged."""
        if arith not in ("f32", "bf16x3"):
            raise ValueError(f"arith must be 'f32' or 'bf16x3' (got {arith!r})")
        self.cfg = cfg
        self.arith = arith
        self.tpitch_align = 32 if arith == "bf16x3" else 4         # transposed-conv workspace rows on 128-byte lines in the bf16x3 mode (conv.tconv_pitch)
        self.plan = SynthesisPlan(sd, cfg, device, arith=arith)
        self.device = self.plan.device
        self.input_shape = [None, cfg.k, cfg.z_dim]
        self.cond_shape = [None, 0]
        self.z_dim, self.w_dim, self.k, self.c_dim = cfg.z_dim, cfg.w_dim, cfg.k, 0
        self.img_resolution, self.img_channels = cfg.img_resolution, cfg.img_channels
        self.num_ws = cfg.num_ws
        self.w_avg = torch.as_tensor(np.asarray(sd["mapping.w_avg"], dtype=np.float32), device=self.device)
        self._ws_cache = {}
        self.taps = None
        self.fuse_torgb = True
        self.skip_fused = {}
        self.fuse_skip_up = os.environ.get("MGF_FUSE_SKIP_UP", "1") != "0"      # tuning hook: 0 = the skip branch's own 2x FIR pass everywhere
        self.last_noise = ("none", None)      # (noise_mode, noises) of the latest synthesis call, read by grad.SynthesisGrad
        self.side = torch.cuda.Stream(device=self.device)
        # MGF_OVERLAP_SKIP=1 forks the skip branch onto the side stream (+0.9 % iterations/s).  Off by default for the same reason
        # as the loss/generator pipeline: concurrent kernels stretch each other -- differently under graph replay than in the eager
        # roofline leg -- so per-kernel durations would no longer agree between bench.py and a rocprofv3 trace.
        self.overlap_skip = os.environ.get("MGF_OVERLAP_SKIP", "0") != "0"
        self._workspaces, self._pins = {}, {}
        self.noise_seed, self._noise_epoch = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, None
        self.noise_state = torch.zeros(2, dtype=torch.int64, device=self.device)          # {stream position, ticket}: advanced on the device
        self._alloc(max_batch)

    # ------------------------------------------------------------------ workspace
    # One workspace (activations, style / demod / value arenas, device job tables) per (batch size, flavour), kept for the life of the
    # generator: hipGraphs captured by the projection engines bake these pointers in, so a call with another batch size must never
    # free them (it used to).  Engines `pin` the workspace they captured; only un-pinned workspaces are ever dropped, and only when
    # the total exceeds MGF_WORKSPACE_GB (default 160 of the 288 GB).
    # Two flavours.  FULL: every layer output of every sample has its own tensor (1.6 GB per sample at 1024^2) -- what the backward pass
    # of gradient mode reads, what `taps` / return_att hand out.  LEAN: the literal loop has no backward pass and only ever needs a block's
    # input, its transposed-conv workspace, its two layer outputs and its skip branch at the same time, so the layer outputs are views of
    # seven arenas sized for the largest user and re-used block after block (0.47 GB per sample: 15 instead of 51 GB at 32 steps per forward).
    _WS_FIELDS = ("n", "lean", "w_buf", "styles", "demods", "vtabs", "noise_rand", "bufs", "img", "const_in", "rgbw", "style_jobs", "n_style_jobs",
                  "max_style_cin", "attn_jobs", "n_attn_jobs", "style_jobs_pl", "attn_jobs_pl", "ws_bytes", "ws_gen")

    def _alloc(self, n, lean=None):
        """Make the workspace of batch size n current (created on first use); lean=None keeps the current flavour."""
        cur = (getattr(self, "n", None), getattr(self, "lean", False))
        lean = cur[1] if lean is None else bool(lean)
        if cur[0] is not None and cur in self._workspaces:
            self._workspaces[cur]["noise_rand"] = self.noise_rand              # lazily allocated: remember it with its workspace
        key = (n, lean)
        ws = self._workspaces.pop(key, None)
        if ws is None:
            self._evict_for(n, lean)
            self._create(n, lean)
            ws = {k: getattr(self, k) for k in self._WS_FIELDS}
        self._workspaces[key] = ws                                             # most recently used last
        for k, v in ws.items():
            setattr(self, k, v)

    def pin(self, key=None):
        """Declare that a captured hipGraph references a workspace (default: the current one).  Returns the key to `unpin` with."""
        key = (self.n, self.lean) if key is None else key
        self._pins[key] = self._pins.get(key, 0) + 1
        return key

    def unpin(self, key):
        if self._pins.get(key, 0) > 0:
            self._pins[key] -= 1

    def _evict_for(self, n, lean):
        budget = float(os.environ.get("MGF_WORKSPACE_GB", "160")) * 2 ** 30
        need = self._workspace_bytes(n, lean)
        total = sum(w["ws_bytes"] for w in self._workspaces.values())
        cur = (getattr(self, "n", None), getattr(self, "lean", False))
        for m in list(self._workspaces):                                       # least recently used first
            if total + need <= budget:
                break
            if self._pins.get(m, 0) == 0 and m != cur:
                total -= self._workspaces.pop(m)["ws_bytes"]

    def _fuses_skip_up(self, l1, n):
        """Does conv1 of this block up-sample the half-resolution skip tensor in its own (form-3 Winograd) epilogue at batch size n?"""
        res = l1.res
        if l1.pcb is not None and res <= BF_DIRECT_MAX_RES:       # bf16x3 mode: this layer takes the direct kernel (full-resolution residual)
            return False
        return bool(self.fuse_skip_up and self.plan.fir_is_1331 and cv.WINOGRAD_FORM == 3 and l1.attn is None and l1.wino_u is not None
                    and l1.wino_u.ndim == 4 and res % 2 == 0 and cv.winograd_fills_chip(n, l1.cout, res, res))

    def _lean_arenas(self, n):
        """Floats per sample of the lean flavour's arenas: T (transposed-conv workspace, also conv_last's output), U (conv0; conv1 before its
        attention), V (conv0 after its attention), W0 / W1 (block outputs, alternating), S (full-resolution skip tensor of the blocks whose
        conv1 does not up-sample it itself), SL (half-resolution skip tensor)."""
        cfg = self.cfg
        conv1 = {lp.res: lp for lp in self.plan.layers if lp.name.endswith(".conv1")}
        a = dict(T=0, U=0, V=0, W0=0, W1=0, S=0, SL=0)
        for i, res in enumerate(cfg.block_resolutions):
            c, px = cfg.channels(res), res * res
            a["U"] = max(a["U"], c * px)
            a["W%d" % (i & 1)] = max(a["W%d" % (i & 1)], c * px)
            if res > 4:
                a["T"] = max(a["T"], c * (res + 1) * cv.tconv_pitch(res // 2, self.tpitch_align))
                a["SL"] = max(a["SL"], c * px // 4)
                if not self._fuses_skip_up(conv1[res], n):
                    a["S"] = max(a["S"], c * px)
                if cfg.has_attention(res):
                    a["V"] = max(a["V"], c * px)
            if res == cfg.img_resolution:
                a["T"] = max(a["T"], c * px)
        return a

    def _workspace_bytes(self, n, lean=False):
        cfg = self.cfg
        if lean:
            return 4 * n * (sum(self._lean_arenas(n).values()) + cfg.img_channels * cfg.img_resolution ** 2)
        per = 0
        for res in cfg.block_resolutions:
            c = cfg.channels(res)
            planes = 1 + (4 if res > 4 else 0) + (1 if res > 4 and cfg.has_attention(res) else 0) + (1 if cfg.has_attention(res) else 0) \
                + (1 if res == cfg.img_resolution else 0)
            per += planes * c * res * res
        return 4 * n * (per + cfg.img_channels * cfg.img_resolution ** 2)

    def _create(self, n, lean=False):
        cfg, dev, P = self.cfg, self.device, self.plan
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        self.n, self.lean = n, bool(lean)
        # generation id: consumers that cache raw pointers into this workspace (grad.GeneratorGrad's job tables) rebuild them when a
        # workspace of the same batch size was evicted and created anew
        self._gen_counter = getattr(self, "_gen_counter", 0) + 1
        self.ws_gen = self._gen_counter
        self.w_buf = e(n, cfg.k, cfg.w_dim)
        self.styles = e(n, P.s_total)
        self.demods = e(n, P.d_total)
        self.vtabs = e(n, max(P.v_total, 1))
        self.noise_rand = None
        self.bufs = {}
        if lean:
            arena = {k: e(n * v) for k, v in self._lean_arenas(n).items() if v}
            conv1 = {lp.res: lp for lp in P.layers if lp.name.endswith(".conv1")}
            view = lambda k, *shape: arena[k][:int(np.prod(shape))].view(*shape)
            for i, res in enumerate(cfg.block_resolutions):
                c, att, W = cfg.channels(res), cfg.has_attention(res), "W%d" % (i & 1)
                b = {"conv1": view("U" if att else W, n, c, res, res)}
                if att:
                    b["conv1a"] = view(W, n, c, res, res)
                if res > 4:
                    b["skip_low"] = view("SL", n, c, res // 2, res // 2)
                    b["t"] = view("T", n, c, res + 1, cv.tconv_pitch(res // 2, self.tpitch_align))
                    b["conv0"] = view("U", n, c, res, res)
                    if att:
                        b["conv0a"] = view("V", n, c, res, res)
                    if not self._fuses_skip_up(conv1[res], n):
                        b["skip"] = view("S", n, c, res, res)
                if res == cfg.img_resolution:
                    b["last"] = view("T", n, c, res, res)
                self.bufs[res] = b
        for res in ([] if lean else cfg.block_resolutions):
            c = cfg.channels(res)
            b = {"conv1": e(n, c, res, res)}
            if res > 4:
                b["skip_low"] = e(n, c, res // 2, res // 2)
                b["skip"] = e(n, c, res, res)
                b["t"] = e(n, c, res + 1, cv.tconv_pitch(res // 2, self.tpitch_align))
                b["conv0"] = e(n, c, res, res)
                if cfg.has_attention(res):
                    b["conv0a"] = e(n, c, res, res)
            if cfg.has_attention(res):
                b["conv1a"] = e(n, c, res, res)
            if res == cfg.img_resolution:
                b["last"] = e(n, c, res, res)
            self.bufs[res] = b
        self.img = e(n, cfg.img_channels, cfg.img_resolution, cfg.img_resolution)
        self.const_in = P.const.unsqueeze(0).repeat(n, 1, 1, 1).contiguous()            # networks.py:1147
        self.rgbw = e(n, cfg.img_channels, cfg.channels(cfg.img_resolution))
        self.ws_bytes = self._workspace_bytes(n, lean)
        self._build_jobs(n)

    def _build_jobs(self, n):
        """Device job tables for the batched style/demod and attention-value launches.  The style / demod / value arenas are
        layer-major: [layer][n][channels], so each job sees a dense [n, c] block.  Two tables each: for one latent set shared by all
        layers (w [n, k, D]: what mapping() broadcasts and every driver uses) and for per-layer latents (ws [n, k, num_ws, D],
        networks.py:1252-1253: layer `slot` reads ws[:, :, slot]) -- same kernels, different offsets into the latent tensor."""
        cfg, P = self.cfg, self.plan
        D, NW = cfg.w_dim, cfg.num_ws

        def tables(per_layer):
            sj = (_lib.StyleJob * len(P.layers))()
            aj = []
            for i, lp in enumerate(P.layers):
                g_off = ((cfg.k - 1) * NW + lp.slot) * D if per_layer else (cfg.k - 1) * D          # the global component (get_global)
                sj[i] = _lib.StyleJob(lp.aff_w.data_ptr(), lp.aff_b.data_ptr(), _lib.ptr(lp.pc.wsq) if lp.demod else 0,
                                      self.styles.data_ptr() + 4 * lp.s_off * n,
                                      (self.demods.data_ptr() + 4 * lp.d_off * n) if lp.demod else 0,
                                      lp.cin, lp.cout, g_off, 1.0 / math.sqrt(D), lp.style_gain)
                if lp.attn is not None:
                    aj.append(_lib.AttnJob(lp.attn.wmv.data_ptr(), lp.attn.bmv.data_ptr(),
                                           self.vtabs.data_ptr() + 4 * lp.attn.v_off * n, lp.attn.c, lp.slot * D if per_layer else 0))
            sjt = torch.frombuffer(bytearray(bytes(sj)), dtype=torch.uint8).to(self.device)
            ajt = None
            if aj:
                arr = (_lib.AttnJob * len(aj))(*aj)
                ajt = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            return sjt, ajt, len(aj)

        self.style_jobs, self.attn_jobs, self.n_attn_jobs = tables(False)
        self.style_jobs_pl, self.attn_jobs_pl, _ = tables(True)
        self.n_style_jobs = len(P.layers)
        self.max_style_cin = max(lp.cin for lp in P.layers)

    def _s(self, lp):
        return self.styles.view(-1)[lp.s_off * self.n:(lp.s_off + lp.cin) * self.n].view(self.n, lp.cin)

    def _d(self, lp):
        return self.demods.view(-1)[lp.d_off * self.n:(lp.d_off + lp.cout) * self.n].view(self.n, lp.cout)

    def _v(self, lp):
        T = self.cfg.k - 1
        a = lp.attn
        return self.vtabs.view(-1)[a.v_off * self.n:(a.v_off + a.c * T) * self.n]

    # ------------------------------------------------------------------ API
    def to(self, *args, **kwargs):
        """nn.Module.to as the drivers use it (`G = ...["Gs"].to(device)`, ...sqz_MSE.py:249).  This generator's weights, packed tables and
        workspaces live on the device it was built on: that device (or float32) is a no-op, anything else RAISES -- a request to move is never
        silently ignored (build the Generator / call loader.load_network with device= instead)."""
        device = kwargs.get("device")
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif isinstance(a, (str, int, torch.device)):
                device = a
            elif isinstance(a, torch.Tensor):
                device, dtype = a.device, a.dtype
        if dtype is not None and dtype != torch.float32:
            raise _lib.MgfError(f"Generator.to: the HIP generator is float32 only (got {dtype})")
        if device is not None:
            want = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
            index = lambda d: d.index if d.index is not None else torch.cuda.current_device()
            if want.type != "cuda" or index(want) != index(self.device):
                raise _lib.MgfError(f"Generator.to: this generator was built on {self.device} (index {index(self.device)}) and cannot move to {want}; "
                                    "construct it on the target device (Generator(sd, cfg, device=...) / loader.load_network(path, device=...))")
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise _lib.MgfError("Generator.train: the HIP generator is inference-only (its weights are constants); use eval()")
        return self

    def requires_grad_(self, flag=False):
        """The reference drivers call .requires_grad_(False); the weights here never carry gradients, so True is refused."""
        if flag:
            raise _lib.MgfError("Generator.requires_grad_(True): the HIP generator's weights are constants (gradient mode differentiates "
                                "with respect to the latent: grad.GeneratorGrad / GradientProjectionEngine)")
        return self

    def mapping(self, z, truncation_psi=1, truncation_cutoff=None):
        """z [n,k,D] -> ws [n,k,num_ws,D], reference MappingNetwork.forward semantics (networks.py:929-941): broadcast over the
        layer slots, then `w_avg.lerp(x, psi)` on all slots or, with a cutoff, on slots [0, cutoff) only."""
        return self._truncate(self._mapping_into(z), truncation_psi, truncation_cutoff)

    def _truncate(self, w, truncation_psi, truncation_cutoff):
        ws = w.unsqueeze(2).expand(-1, -1, self.num_ws, -1)
        if truncation_psi != 1:
            if truncation_cutoff is None:
                ws = self.w_avg.lerp(ws, truncation_psi)
            else:
                ws = torch.cat([self.w_avg.lerp(ws[:, :, :truncation_cutoff], truncation_psi), ws[:, :, truncation_cutoff:]], dim=2)
        return ws

    def _mapping_into(self, z):
        _lib.require_gpu(z)
        n = z.shape[0]
        assert tuple(z.shape[1:]) == (self.cfg.k, self.cfg.z_dim), z.shape
        if n != self.n:
            self._alloc(n)
        z = z.contiguous().float()
        save = getattr(self, "map_save", None)       # gradient mode (grad.GeneratorGrad) keeps the mapping activations for its backward
        if save is not None:
            rc = _lib.lib().mgf_mapping_forward_save(self.w_buf.data_ptr(), z.data_ptr(), self.plan.mapping_blob.data_ptr(),
                                                     save.data_ptr(), n, self.cfg.k, self.cfg.w_dim, self.cfg.mapping_layers // 2,
                                                     int(self.cfg.normalize_global), _lib.stream_ptr())
        else:
            rc = _lib.lib().mgf_mapping_forward(self.w_buf.data_ptr(), z.data_ptr(), self.plan.mapping_blob.data_ptr(), n,
                                                self.cfg.k, self.cfg.w_dim, self.cfg.mapping_layers // 2,
                                                int(self.cfg.normalize_global), _lib.stream_ptr())
        _lib.check(rc, "mapping_forward")
        return self.w_buf

    def synthesis(self, w, noise_mode="random", noises=None, return_att=False):
        """w: [n, k, D] -- one latent set shared by all layers, as every driver uses it -- or ws [n, k, num_ws, D] with per-layer
        latents (W+; layer `slot` reads ws[:, :, slot] for its style AND its attention values, networks.py:1022-1031,1252-1253).
        Returns img [n,3,R,R] (the workspace buffer: valid until the next call with this batch size)."""
        cfg, P, L = self.cfg, self.plan, _lib.lib()
        n = w.shape[0]
        if n != self.n:
            raise _lib.MgfError("synthesis: batch size changed; call through __call__ / mapping first")
        if self.lean and (self.taps is not None or return_att):
            # the lean flavour's layer outputs are views of shared arenas that later blocks overwrite: tensors that are handed out need their
            # own storage (forward_workspace makes the same choice before it gets here; this is the direct-call path)
            self._alloc(n, False)
        st = _lib.stream_ptr()
        D, T = cfg.w_dim, cfg.k - 1
        _lib.require_gpu(w)
        per_layer = w.ndim == 4
        if per_layer:
            if tuple(w.shape) != (n, cfg.k, cfg.num_ws, D):
                raise _lib.MgfError(f"synthesis: ws must be [n, {cfg.k}, {cfg.num_ws}, {D}] (got {tuple(w.shape)})")
            w = w.contiguous().float()
            stride_n, stride_t = cfg.k * cfg.num_ws * D, cfg.num_ws * D
            sjobs, ajobs = self.style_jobs_pl, self.attn_jobs_pl
        else:
            assert w.is_contiguous() and tuple(w.shape) == (n, cfg.k, D)
            stride_n, stride_t = cfg.k * D, D
            sjobs, ajobs = self.style_jobs, self.attn_jobs
        self.last_w = w                                                            # kept alive for the asynchronous launches below
        _lib.check(L.mgf_style_demod_multi(sjobs.data_ptr(), self.n_style_jobs, w.data_ptr(), stride_n, n, D, self.max_style_cin, st),
                   "style_demod_multi")
        if self.n_attn_jobs:
            _lib.check(L.mgf_attn_values_multi(ajobs.data_ptr(), self.n_attn_jobs, w.data_ptr(), stride_n, stride_t, n, T, D, st),
                       "attn_values_multi")
        if noise_mode == "random":
            noises = self._draw_noise(n)
        self.last_noise = (noise_mode, noises)
        layers = {lp.name: lp for lp in P.layers}
        self.att_maps = {} if return_att else None
        x = None
        for res in cfg.block_resolutions:
            b = f"synthesis.b{res}"
            B = self.bufs[res]
            if res == 4:
                x_in = self.const_in
                x = self._layer(layers[b + ".conv1"], x_in, B, "conv1", noise_mode, noises, residual=None)
            else:
                # the resnet skip branch (1x1 conv + 2x FIR upsample: memory bound) runs on a side stream next to conv0's
                # MFMA-bound transposed conv; both only read x, and conv1 joins them
                l1 = layers[b + ".conv1"]
                # conv1 on the form-3 Winograd kernel without attention (256^2 and larger): its epilogue up-samples the half-resolution
                # skip output itself -- no fir_up2 pass, no full-resolution skip tensor (one write + one read of the block's largest map)
                fuse_up = self._fuses_skip_up(l1, n)
                if not fuse_up and "skip" not in B:           # lean workspace: laid out with the same predicate, for the flag's value then
                    raise _lib.MgfError(f"synthesis: the lean workspace has no full-resolution skip tensor for the {res}x{res} block "
                                        "(fuse_skip_up changed after the workspace was created): use the full workspace")
                self.skip_fused[res] = bool(fuse_up)         # (gradient mode's backward reads the skip tensor at the resolution it was consumed)
                main = torch.cuda.current_stream(self.device)
                if self.overlap_skip:
                    self.side.wait_stream(main)
                    with torch.cuda.stream(self.side):
                        cv.conv_forward(x, P.skips[res], out=B["skip_low"])
                        if not fuse_up:
                            cv.upfirdn_into(B["skip"], B["skip_low"], P.fir, up=2, pad=(2, 1, 2, 1), gain=4.0, separable=True)
                else:
                    cv.conv_forward(x, P.skips[res], out=B["skip_low"])
                    if not fuse_up:
                        cv.upfirdn_into(B["skip"], B["skip_low"], P.fir, up=2, pad=(2, 1, 2, 1), gain=4.0, separable=True)
                x0 = self._layer(layers[b + ".conv0"], x, B, "conv0", noise_mode, noises, residual=None)
                if self.overlap_skip:
                    main.wait_stream(self.side)
                if fuse_up:
                    x = self._layer(l1, x0, B, "conv1", noise_mode, noises, residual=None, residual_low=B["skip_low"])
                else:
                    x = self._layer(l1, x0, B, "conv1", noise_mode, noises, residual=B["skip"])
            if self.taps is not None:
                self.taps[b] = x
            if res == cfg.img_resolution:
                lp = layers[b + ".conv_last"]
                lt = layers[b + ".torgb"]
                if self.taps is None and self.fuse_torgb and lp.cout <= 32:
                    # conv_last + ToRGB in one kernel: the [n,32,R,R] conv_last activation never goes to HBM
                    _lib.check(L.mgf_rgb_weights_f32(self.rgbw.data_ptr(), lt.w_raw.data_ptr(), self._s(lt).data_ptr(), n, lt.w_raw.shape[0],
                                                     lt.w_raw.shape[1], st), "rgb_weights")                       # W[c,co] * s[n,co]
                    if lp.wino_u is not None and lp.wino_u.ndim == 4 and lp.cout == 32 and cv.winograd_fills_chip(n, lp.cout, res, res):
                        cv.winograd2_rgb_forward(x, lp.wino_u, self.rgbw, lt.bias, self.img, in_scale=self._s(lp), out_scale=self._d(lp))
                    else:
                        cv.conv_forward(x, lp.pc, pad=(1, 1), in_scale=self._s(lp), out_scale=self._d(lp),
                                        rgb=(self.rgbw, lt.bias, self.img))
                else:
                    if lp.wino_u is not None and cv.winograd_fills_chip(n, lp.cout, res, res):
                        x = cv.winograd_forward(x, lp.wino_u, in_scale=self._s(lp), out_scale=self._d(lp), out=B["last"])
                    else:
                        x = cv.conv_forward(x, lp.pc, pad=(1, 1), in_scale=self._s(lp), out_scale=self._d(lp), out=B["last"])
                    if self.taps is not None:
                        self.taps[b] = x
                    ep = _lib.make_epilogue(bias=lt.bias)
                    cv.conv_forward(x, lt.pc, in_scale=self._s(lt), epilogue=ep, out=self.img)
        return self.img

    def list2tensor(self, att_maps=None):
        """SynthesisNetwork.list2tensor (networks.py:1222-1242): the per-layer attention maps of the latest return_att=True call,
        nearest-neighbour replicated to the image resolution and stacked -> [n, k-1, layers, 1, R, R] (738 MB at 1024^2, n=1;
        the projection drivers discard it, so it is only built on request)."""
        att_maps = self.att_maps if att_maps is None else att_maps
        if not att_maps:
            return torch.zeros([1], device=self.device)
        T, R = self.cfg.k - 1, self.cfg.img_resolution
        names = [lp.name for lp in self.plan.layers if lp.name in att_maps]
        n = att_maps[names[0]][0].shape[0]
        out = torch.empty([n, T, len(names), 1, R, R], dtype=torch.float32, device=self.device)
        for i, name in enumerate(names):
            probs = att_maps[name][0]
            side = math.isqrt(probs.shape[1])
            _lib.check(_lib.lib().mgf_att_map_upsample_f32(out.data_ptr(), probs.data_ptr(), n, side, T, R, i, len(names),
                                                           _lib.stream_ptr()), "att_map_upsample")
        return out

    def _noise_for(self, lp, noise_mode, noises):
        if lp.noise_strength is None or noise_mode == "none":
            return None, 1
        if noise_mode == "const":
            return lp.noise_const, 1
        t = noises[lp.name]
        return t, t.shape[0]

    def seed_noise(self, seed: int):
        """Restart the per-layer noise stream of noise_mode="random" under an explicit key, independent of torch's global CUDA generator
        (see _draw_noise for how torch.manual_seed governs it otherwise).  Captured launch sequences read the key at capture time and the
        stream position from the device: seed before capture; replays then continue the device-side stream."""
        self.noise_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.noise_state.zero_()
        gen = torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()]
        self._noise_epoch = (int(gen.initial_seed()), int(gen.get_offset()))       # "nothing happened since": the next draw keeps this key

    def _draw_noise(self, n):
        """Fresh N(0,1) maps per layer and call (networks.py:1016-1017): ONE mgf_randn_f32 launch fills a flat buffer that is
        laid out layer-major ([layer][n][r*r]), so every layer sees a dense [n, r, r] view without copies.

        Relation to torch's global CUDA generator, chosen to match what the reference's `torch.randn` calls do to it: a draw outside graph
        capture advances the global offset (as torch.randn would: later torch draws shift, like after the reference's forward), and
        `torch.manual_seed(s)` -- or any foreign draw -- in front of a call re-keys the stream, so `manual_seed(s); G(z)` is reproducible
        call for call.  Replays of a captured graph cannot see the host generator: they continue the device-side stream of the key the
        capture saw (an eager loop and a replayed loop of one engine therefore draw different, equally distributed noise).  `seed_noise`
        gives the stream a key of its own."""
        sizes = [(lp.name, lp.res) for lp in self.plan.layers if lp.noise_strength is not None]
        total = sum(r * r for _, r in sizes)
        if self.noise_rand is None or self.noise_rand.numel() != n * total:
            self.noise_rand = torch.empty(n * total, dtype=torch.float32, device=self.device)
        # The library's own Philox / Box-Muller kernel: its stream position lives on the device and every launch -- every replay of a
        # captured graph -- advances it.  torch.manual_seed still governs it, like the reference's torch.randn: outside graph capture the
        # call looks at torch's CUDA generator -- (seed, offset) other than what the previous draw left behind means the user re-seeded (or
        # drew elsewhere): the stream restarts under a key derived from that pair -- and leaves its own mark (offset + 4) behind.
        if not torch.cuda.is_current_stream_capturing():
            gen = torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()]
            seed, off = int(gen.initial_seed()), int(gen.get_offset())
            if (seed, off) != self._noise_epoch:
                self.noise_seed = (seed ^ (off * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
                self.noise_state.zero_()
            gen.set_offset(off + 4)
            self._noise_epoch = (seed, off + 4)
        _lib.check(_lib.lib().mgf_randn_f32(self.noise_rand.data_ptr(), self.noise_rand.numel(), self.noise_seed, self.noise_state.data_ptr(),
                                            _lib.stream_ptr()), "randn")
        out, off = {}, 0
        for name, r in sizes:
            out[name] = self.noise_rand[off * n:(off + r * r) * n].view(n, r * r)
            off += r * r
        return out

    def _bf_direct(self, lp, n, residual_low):
        """Does this 3x3 layer take the bf16x3 direct kernel?  Only in that mode, only where it wins: maps of 32^2 .. 128^2 (the layers whose
        Winograd launch has no fused skip up-sampling to lose), and enough tiles to fill the chip without split-K."""
        return (lp.pcb is not None and lp.kind == "conv3" and residual_low is None and 32 <= lp.res <= BF_DIRECT_MAX_RES
                and n * (lp.res // 8) * (lp.res // 32) * (lp.cout // 32) >= 512)

    def _layer(self, lp, x, B, key, noise_mode, noises, residual, residual_low=None):
        """One SynthesisLayer (networks.py:1010-1042): modulated conv (+FIR) -> [attention] -> noise -> bias/lrelu."""
        n = x.shape[0]
        noise, noise_n = self._noise_for(lp, noise_mode, noises)
        ep = _lib.make_epilogue(bias=lp.bias, noise=noise, noise_strength=lp.noise_strength if noise is not None else None,
                                noise_n=noise_n, act="lrelu", alpha=0.2, gain=lp.act_gain, residual=residual)
        has_att = lp.attn is not None
        s, d = self._s(lp), self._d(lp)
        if lp.kind == "tconv":
            t = cv.tconv3x3s2_forward(x, lp.pc, in_scale=s, out_scale=d, out=B["t"], bf=lp.pcb)
            # plan.fir is the outer product of the 1-D resample kernel (networks.py:1113 / upfirdn2d.setup_filter): separable
            y = cv.upfirdn_into(B[key], t, self.plan.fir, up=1, pad=(1, 1, 1, 1), gain=4.0, epilogue=None if has_att else ep,
                                separable=True)
        elif self._bf_direct(lp, n, residual_low):
            # bf16x3 mode: the direct 3x3 form at three bf16 matrix instructions per 16 channels beats float32 Winograd below 256^2
            y = cv.conv_forward(x, lp.pc, pad=(1, 1), in_scale=s, out_scale=d, epilogue=None if has_att else ep, out=B[key], bf=lp.pcb)
        elif lp.wino_u is not None and cv.winograd_fills_chip(n, lp.cout, lp.res, lp.res):
            y = cv.winograd_forward(x, lp.wino_u, in_scale=s, out_scale=d, epilogue=None if has_att else ep, out=B[key],
                                    residual_low=residual_low)
        else:
            y = cv.conv_forward(x, lp.pc, pad=(1, 1), in_scale=s, out_scale=d, epilogue=None if has_att else ep, out=B[key])
        if self.taps is not None:
            self.taps[lp.name + ":conv"] = y
        if has_att:
            a = lp.attn
            out = B[key + "a"]
            probs = argmax = None
            if self.att_maps is not None:
                probs = torch.empty([n, a.f, self.cfg.k - 1], dtype=torch.float32, device=self.device)
                argmax = torch.empty([n, a.f], dtype=torch.int32, device=self.device)
                self.att_maps[lp.name] = (probs, argmax)
            rc = _lib.lib().mgf_duplex_attention(out.data_ptr(), y.data_ptr(), a.wqc.data_ptr(), a.spos.data_ptr(),
                                                 self._v(lp).data_ptr(), n, a.c, a.f, self.cfg.k - 1, C.byref(ep), lp.res,
                                                 _lib.ptr(probs), _lib.ptr(argmax), _lib.stream_ptr())
            _lib.check(rc, "duplex_attention")
            y = out
        return y

    def forward_workspace(self, z=None, c=None, ws=None, truncation_psi=1, truncation_cutoff=None, return_img=True, return_att=False,
                          return_ws=False, subnet=None, noise_mode="random", noises=None, fused_modconv=None, att_format="tensor", lean=False):
        """Generator.forward (networks.py:1304-1331), zero-copy: the returned image IS the workspace buffer of this batch size and is
        overwritten by the next call (what the projection engines want; `__call__` hands out copies).

        ws [n, k, num_ws, D]: per-layer latents are honoured (every layer reads its own slot, :1252-1253).  truncation_psi /
        truncation_cutoff act in the mapping network only, i.e. when `z` is given (:1317, :935-941).  return_att=True returns the
        stacked attention tensor [n, k-1, layers, 1, R, R] of list2tensor (:1262,1222-1242); att_format="maps" returns the per-layer
        dict {layer: (probs [n,F,k-1], argmax [n,F])} instead (the cheap form: the stacked tensor is 738 MB per image at 1024^2)."""
        lean = bool(lean) and self.taps is None and not return_att      # (intermediate tensors are handed out: they need their own storage)
        nb = (z if ws is None else ws)
        if nb is not None and (nb.shape[0], lean) != (self.n, self.lean):
            self._alloc(nb.shape[0], lean)
        return_tensor = False
        if subnet is not None:
            return_ws, return_img, return_att, return_tensor = subnet == "mapping", subnet == "synthesis", False, True
        if att_format not in ("tensor", "maps"):
            raise ValueError(f"att_format must be 'tensor' or 'maps' (got {att_format!r})")
        ws_out = None
        if ws is None:
            if z is None:
                raise _lib.MgfError("Generator: pass z or ws")
            w = self._mapping_into(z)
            if truncation_psi != 1 and truncation_cutoff is not None:
                w_in = ws_out = self._truncate(w, truncation_psi, truncation_cutoff).contiguous()    # slots differ -> per-layer path
            else:
                w_in = self.w_avg.lerp(w, truncation_psi) if truncation_psi != 1 else w
        else:
            _lib.require_gpu(ws)
            if ws.ndim != 4 or tuple(ws.shape[1:]) != (self.cfg.k, self.num_ws, self.cfg.w_dim):
                raise _lib.MgfError(f"Generator: ws must be [n, {self.cfg.k}, {self.num_ws}, {self.cfg.w_dim}] (got {tuple(ws.shape)})")
            if ws.shape[0] != self.n:
                self._alloc(ws.shape[0])
            w_in = ws_out = ws.contiguous().float()
        ret = ()
        if return_img or return_att:
            img = self.synthesis(w_in if w_in.ndim == 4 else w_in.contiguous(), noise_mode=noise_mode, noises=noises, return_att=return_att)
            if return_img:
                ret += (img,)
            if return_att:
                ret += (self.att_maps if att_format == "maps" else self.list2tensor(self.att_maps),)
        if return_ws:
            ret += (ws_out if ws_out is not None else w_in.unsqueeze(2).expand(-1, -1, self.num_ws, -1),)
        return ret[0] if return_tensor else ret

    def __call__(self, *args, **kwargs):
        """Generator.forward with the reference's ownership: every returned tensor is the caller's own (`a = G(z1)[0]; b = G(z2)[0]` do
        not alias).  The projection engines use `forward_workspace` (no copy)."""
        ret = self.forward_workspace(*args, **kwargs)
        own = lambda t: t.clone() if isinstance(t, torch.Tensor) else t
        return own(ret) if not isinstance(ret, tuple) else tuple(own(t) for t in ret)
