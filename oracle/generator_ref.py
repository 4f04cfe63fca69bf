"""ORACLE (test infrastructure only) -- CPU restatement of the GANformer generator forward.

Functional (state-dict driven) float32 torch-CPU restatement of training/networks.py,
written from SURVEY.md appendix A; every step cites the reference lines it follows.  It
executes the *reference's* order of operations (per-sample modulated weights, the full
query projection, P.V then the modulation FC, ...) so that it is an independent check of
the algebraically re-associated HIP engine.  Dead work (key projection, Q.K^T,
carried-assignment centroids, list2tensor) is not executed.

Pinned against the reference module itself by tests/golden/gen_*.npz
(oracle/make_golden.py; container only).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .ops_ref import bias_act_ref, conv2d_resample_ref, modulated_conv2d_ref, upfirdn2d_ref

SQRT2 = math.sqrt(2.0)
SQRT_HALF = math.sqrt(0.5)


def to_torch_state(sd):
    return {k: torch.as_tensor(v, dtype=torch.float32) for k, v in sd.items()}


def _fc(sd, prefix, x, lrmul=1.0, act="linear"):
    """FullyConnectedLayer (networks.py:131-150): y = x (w g)^T + b g_b; lrelu -> *sqrt2."""
    w = sd[prefix + ".weight"]
    g = lrmul / math.sqrt(w.shape[1])
    y = x.matmul((w * g).t())
    b = sd.get(prefix + ".bias")
    if b is not None:
        y = y + b * lrmul
    if act == "lrelu":
        y = F.leaky_relu(y, 0.2) * SQRT2
    return y


def _normalize(x, eps=1e-8):
    """networks.py:30-37: joint second moment over all non-batch dims."""
    dims = list(range(1, x.ndim))
    return x * (x.square().mean(dim=dims, keepdim=True) + eps).rsqrt()


def _latent_self_attention(sd, prefix, x, pos, T):
    """Mapping-network TransformerLayer (integration 'add', no norm, no k-means): networks.py:748-822."""
    D = x.shape[-1]
    B = x.shape[0] // T
    q = _fc(sd, prefix + ".to_queries", x) + _fc(sd, prefix + ".from_pos_map", pos).repeat(B, 1)
    k = _fc(sd, prefix + ".to_keys", x) + _fc(sd, prefix + ".to_pos_map", pos).repeat(B, 1)
    v = _fc(sd, prefix + ".to_values", x)
    q, k, v = (t.reshape(B, T, D) for t in (q, k, v))
    scores = q.matmul(k.transpose(1, 2)) / math.sqrt(D)
    probs = torch.softmax(scores, dim=-1)
    ctl = probs.matmul(v).reshape(B * T, D)
    return x + _fc(sd, prefix + ".modulation", ctl)


def mapping_ref(sd, z, cfg):
    """MappingNetwork.forward (networks.py:894-942) -> w [B, k, w_dim] (identical for all num_ws slots)."""
    B = z.shape[0]
    T = cfg.k - 1
    lr = cfg.mapping_lrmul
    n_res = cfg.mapping_layers // 2
    zl, g = z[:, :T].float(), z[:, T:].float()
    if cfg.normalize_global:
        g = _normalize(g)
    zl = _normalize(zl)

    def mlp(prefix, x, sa):
        for i in range(n_res):
            x_in = x
            if sa:
                x = _latent_self_attention(sd, f"{prefix}.sa{i}", x, sd["pos"], T)
            h = _fc(sd, f"{prefix}.l{i}.fc0", x, lr, "lrelu")
            h = _fc(sd, f"{prefix}.l{i}.fc1", h, lr)
            x = F.leaky_relu(h + x_in, 0.2)
        return _fc(sd, prefix + ".out_layer", x, lr, "lrelu")

    gw = mlp("mapping.global_mlp", g.reshape(B, -1), False).reshape(B, 1, -1)
    lw = mlp("mapping.mlp", zl.reshape(B * T, -1), True).reshape(B, T, -1)
    return torch.cat([lw, gw], dim=1)


def duplex_attention_ref(sd, prefix, x, y_comp, grid_pos, return_probs=False):
    """Image<-latents attention of a SynthesisLayer (networks.py:748-822 with kmeans, parametric centroids,
    integration 'mul', layer norm).  x: [B,C,r,r]; y_comp: [B,T,D]."""
    B, C, H, W = x.shape
    Fn = H * W
    T = y_comp.shape[1]
    X = x.reshape(B, C, Fn).permute(0, 2, 1).reshape(B * Fn, C)                       # :1028
    q = _fc(sd, prefix + ".to_queries", X)                                             # :757
    qp = q + _fc(sd, prefix + ".from_pos_map", grid_pos.reshape(Fn, -1)).repeat(B, 1)  # :763-764
    e = torch.cat([q, qp - q], dim=-1).reshape(B, Fn, 2 * C)                            # :688
    cent = sd[prefix + ".centroids"].reshape(T, 2 * C)                                 # :715-717
    aw = sd[prefix + ".att_weight"].reshape(1, 1, 2 * C)
    scores = (e * aw).matmul(cent.t()) / math.sqrt(C)                                  # :792-795
    probs = torch.softmax(scores, dim=-1)                                              # :801
    v = _fc(sd, prefix + ".to_values", y_comp.reshape(B * T, -1)).reshape(B, T, C)     # :759
    ctl = probs.matmul(v).reshape(B * Fn, C)                                           # :812-814
    Xn = X.reshape(B, Fn, C)
    Xn = Xn * torch.rsqrt(Xn.square().mean(dim=2, keepdim=True) + 1e-8)               # :349-354
    out = Xn * (_fc(sd, prefix + ".modulation", ctl).reshape(B, Fn, C) + 1)            # :662-668
    out = out.permute(0, 2, 1).reshape(B, C, H, W)                                     # :1034
    if return_probs:
        return out, probs
    return out


def synthesis_layer_ref(sd, prefix, x, w, cfg, res, up, noise, gain=1.0, taps=None):
    """SynthesisLayer.forward (networks.py:1010-1042).  w: [B,k,D]; noise: [*,res,res] tensor or None."""
    styles = _fc(sd, prefix + ".affine", w[:, -1])                                     # :1022
    weight = sd[prefix + ".weight"]
    wg = 1.0 / math.sqrt(weight.shape[1] * weight.shape[2] * weight.shape[3])
    x = modulated_conv2d_ref(x, weight * wg, styles, up=up, padding=1,
                             resample_kernel=sd[prefix + ".resample_kernel"], flip_weight=(up == 1))
    if taps is not None:
        taps[prefix + ":conv"] = x
    if (prefix + ".transformer.to_queries.weight") in sd:
        x, probs = duplex_attention_ref(sd, prefix + ".transformer", x, w[:, :-1], sd[prefix + ".grid_pos"], True)
        if taps is not None:
            taps[prefix + ":probs"] = probs
    if (prefix + ".noise_strength") in sd and noise is not None:
        x = x + noise.reshape(-1, 1, res, res) * sd[prefix + ".noise_strength"]        # :1015-1020,1036
    if (prefix + ".biasAct.bias") in sd:
        x = bias_act_ref(x, sd[prefix + ".biasAct.bias"], act="lrelu", gain=SQRT2 * gain)  # :1039-1040
    return x


def synthesis_ref(sd, w, cfg, noise_mode="const", noises=None, taps=None):
    """SynthesisNetwork.forward (networks.py:1244-1264) for the resnet architecture; returns img [B,3,R,R].

    noise_mode: 'const' uses the stored noise_const buffers, 'none' disables noise, 'inject' takes
    `noises[prefix]` tensors of shape [B,res,res] (stands in for the reference's fresh randn, :1016-1017).
    """
    B = w.shape[0]
    x = None
    slots = {f"synthesis.b{res}.{name}": slot for res, name, _ci, _co, _up, slot, *_ in cfg.layer_table()}
    if w.ndim == 4:
        # per-layer latents ws [B,k,num_ws,D]: block r sees ws.narrow(2, w_idx, ...) and hands its layers next(w_iter)
        # (networks.py:1134,1158-1173,1249-1253) -> layer `slot` reads ws[:, :, slot]
        ws4, w = w, None
        pick = lambda prefix: ws4[:, :, slots[prefix]]
    else:
        w3 = w
        pick = lambda prefix: w3
    for res in cfg.block_resolutions:
        b = f"synthesis.b{res}"

        def nz(name):
            if noise_mode == "none":
                return None
            if noise_mode == "const":
                return sd[f"{b}.{name}.noise_const"]
            return noises[f"{b}.{name}"]

        if res == 4:
            x = sd[b + ".const"][None].repeat(B, 1, 1, 1)                              # :1147
            x = synthesis_layer_ref(sd, b + ".conv1", x, pick(b + ".conv1"), cfg, res, 1, nz("conv1"), 1.0, taps)
        else:
            ws_ = sd[b + ".skip.weight"]
            y = conv2d_resample_ref(x, ws_ * (1.0 / math.sqrt(ws_.shape[1])), f=sd[b + ".skip.resample_kernel"],
                                    up=2, padding=0, flip_weight=False)                # :245-250
            y = y * SQRT_HALF
            x = synthesis_layer_ref(sd, b + ".conv0", x, pick(b + ".conv0"), cfg, res, 2, nz("conv0"), 1.0, taps)
            x = synthesis_layer_ref(sd, b + ".conv1", x, pick(b + ".conv1"), cfg, res, 1, nz("conv1"), SQRT_HALF, taps)
            x = y + x                                                                  # :1160
        if res == cfg.img_resolution:
            x = synthesis_layer_ref(sd, b + ".conv_last", x, pick(b + ".conv_last"), cfg, res, 1, None, 1.0, taps)   # :1170
        if taps is not None:
            taps[b] = x                       # what SynthesisBlock.forward returns as x (:1174)
        if res == cfg.img_resolution:
            tw = sd[b + ".torgb.weight"]
            styles = _fc(sd, b + ".torgb.affine", pick(b + ".torgb")[:, -1]) * (1.0 / math.sqrt(tw.shape[1]))    # :1056-1059
            img = modulated_conv2d_ref(x, tw, styles, demodulate=False)
            img = bias_act_ref(img, sd[b + ".torgb.biasAct.bias"])
            return img
    raise AssertionError("unreachable")


def truncate_ref(sd, w, cfg, truncation_psi=1, truncation_cutoff=None):
    """Broadcast + truncation of MappingNetwork.forward (networks.py:929-941): w [B,k,D] -> ws [B,k,num_ws,D]."""
    ws = w.unsqueeze(2).repeat(1, 1, cfg.num_ws, 1)
    if truncation_psi != 1:
        if truncation_cutoff is None:
            ws = sd["mapping.w_avg"].lerp(ws, truncation_psi)
        else:
            ws[:, :, :truncation_cutoff] = sd["mapping.w_avg"].lerp(ws[:, :, :truncation_cutoff], truncation_psi)
    return ws


def list2tensor_ref(probs_by_layer, cfg):
    """SynthesisNetwork.list2tensor (networks.py:1222-1242): the layers' attention maps [B,F,T] (in network order), each
    nearest-neighbour replicated to the image resolution, stacked to [B, T, layers, 1, R, R]."""
    R, T = cfg.img_resolution, cfg.k - 1
    out = []
    for p in probs_by_layer:
        B, Fn, _ = p.shape
        r = math.isqrt(Fn)
        m = p.reshape(B, r, r, T).permute(0, 3, 1, 2)                                  # [B,T,r,r]
        out.append(m.repeat_interleave(R // r, dim=2).repeat_interleave(R // r, dim=3))
    return torch.stack(out, dim=2).unsqueeze(3)


def generator_ref(sd, z, cfg, noise_mode="const", noises=None, taps=None, truncation_psi=1, truncation_cutoff=None):
    """Generator.forward(z)[0] (networks.py:1304-1331); truncation_psi=1 in the projection drivers."""
    w = mapping_ref(sd, z, cfg)
    if taps is not None:
        taps["ws"] = w
    if truncation_psi != 1:
        w = truncate_ref(sd, w, cfg, truncation_psi, truncation_cutoff)
    return synthesis_ref(sd, w, cfg, noise_mode, noises, taps)
