"""Deterministic synthetic GANformer generator weights (pure NumPy).

No trained checkpoint exists offline, so parity and benchmarks are
pinned on seeded synthetic weights.  The dictionary produced here uses the reference
``Generator.state_dict()`` key names and shapes (training/networks.py:1269-1302 and the
layers it builds; probed key list in SURVEY.md section 8c) so the same tensors load into the
reference module (oracle/make_golden.py, container only), into the CPU oracle and into
the HIP engine.

Stored-parameter convention (training/networks.py:69-89): the runtime weight is
``stored * gain / sqrt(fan_in) * lrmul`` and the runtime bias is ``stored * lrmul``.  The
reference initialises stored weights with std ``lrmul`` (a defect);
we draw stored weights with std ``1 / lrmul`` so the *effective* weights are He-scaled and
the mapping network is not degenerate.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field

import numpy as np


@dataclass(frozen=True)
class GeneratorConfig:
    """Architecture constants of the GANformer generator (run_network.py:61-77,243-283)."""
    img_resolution: int = 1024
    img_channels: int = 3
    z_dim: int = 32
    w_dim: int = 32
    k: int = 17
    channel_base: int = 32 << 10
    channel_max: int = 512
    attn_max_log2res: int = 8        # attention where log2(res) < end_res (networks.py:1212)
    mapping_layers: int = 8          # -> 4 ResnetLayers + out layer (networks.py:183)
    mapping_lrmul: float = 0.01
    normalize_global: bool = True    # False for TF-converted pickles (loader.py:124)

    @property
    def block_resolutions(self):
        return [2 ** i for i in range(2, int(math.log2(self.img_resolution)) + 1)]

    def channels(self, res: int) -> int:
        return min(self.channel_base // res, self.channel_max)

    def has_attention(self, res: int) -> bool:
        return math.log2(res) < self.attn_max_log2res

    @property
    def num_ws(self) -> int:
        # one conv for the stem, two per later block, + conv_last + torgb on the last block
        return 1 + 2 * (len(self.block_resolutions) - 1) + 2

    @property
    def num_components(self) -> int:
        return self.k - 1

    def conv_gflop(self):
        """Algorithmic GFLOP of the convolutions of ONE generator forward: 2*taps*cin*cout per output
        pixel, the stride-2 transposed conv of conv0 counted per INPUT pixel, plus the 1x1 skip convs at the lower resolution."""
        rows = self.layer_table()
        gf = sum(2 * (9 if nm != "torgb" else 1) * ci * co * ((res // up) ** 2) for res, nm, ci, co, up, *_ in rows)
        gf += sum(2 * self.channels(r // 2) * self.channels(r) * (r // 2) ** 2 for r in self.block_resolutions[1:])
        return gf / 1e9

    def layer_table(self):
        """[(block_res, layer_name, cin, cout, up, ws_slot, has_attention, has_noise_bias)] in execution order."""
        rows = []
        slot = 0
        for res in self.block_resolutions:
            cout = self.channels(res)
            cin = self.channels(res // 2) if res > 4 else cout
            att = self.has_attention(res)
            if res > 4:
                rows.append((res, "conv0", cin, cout, 2, slot, att, True)); slot += 1
            rows.append((res, "conv1", cout, cout, 1, slot, att, True)); slot += 1
            if res == self.img_resolution:
                rows.append((res, "conv_last", cout, cout, 1, slot, False, False)); slot += 1
                rows.append((res, "torgb", cout, self.img_channels, 1, slot, False, False)); slot += 1
        return rows


TINY = GeneratorConfig(img_resolution=64, channel_base=512, channel_max=32)
SMALL256 = GeneratorConfig(img_resolution=256)
FULL1024 = GeneratorConfig(img_resolution=1024)


def sinusoidal_grid(res: int, dim: int) -> np.ndarray:
    """[res,res,dim] positional grid = cat(sinX, cosX, sinY, cosY) (networks.py:406-420), float32 math."""
    c = np.linspace(-1.0, 1.0, res, dtype=np.float32)[:, None]
    i = np.arange(dim // 4, dtype=np.float32)
    denom = np.power(np.float32(10000.0), (np.float32(4.0) * i / np.float32(dim))).astype(np.float32)
    arg = (c / denom).astype(np.float32)
    s, co = np.sin(arg).astype(np.float32), np.cos(arg).astype(np.float32)
    sx = np.broadcast_to(s[None, :, :], (res, res, dim // 4))
    cx = np.broadcast_to(co[None, :, :], (res, res, dim // 4))
    sy = np.broadcast_to(s[:, None, :], (res, res, dim // 4))
    cy = np.broadcast_to(co[:, None, :], (res, res, dim // 4))
    return np.ascontiguousarray(np.concatenate([sx, cx, sy, cy], axis=-1), dtype=np.float32)


def fir_kernel_1331() -> np.ndarray:
    f = np.array([1.0, 3.0, 3.0, 1.0], dtype=np.float32)
    f2 = np.outer(f, f)
    return (f2 / f2.sum()).astype(np.float32)


def make_state_dict(cfg: GeneratorConfig, seed: int = 0, noise_strength: float = 0.1) -> "OrderedDict[str, np.ndarray]":
    """Seeded float32 parameters/buffers under the reference state_dict names."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    D = cfg.w_dim
    T = cfg.num_components

    def normal(*shape, std=1.0):
        return (rng.standard_normal(shape) * std).astype(np.float32)

    def fc(prefix, cin, cout, lrmul=1.0, bias_mean=0.0, bias_std=0.1):
        sd[prefix + ".weight"] = normal(cout, cin, std=1.0 / lrmul)
        sd[prefix + ".bias"] = ((bias_mean + rng.standard_normal(cout) * bias_std) / lrmul).astype(np.float32)

    def transformer(prefix, dim, pos_dim, to_dim, kmeans):
        if kmeans:
            sd[prefix + ".att_weight"] = (1.0 + 0.1 * rng.standard_normal((1, 1, 2 * dim))).astype(np.float32)
            sd[prefix + ".centroids"] = normal(1, 1, T, 2 * dim)
        fc(prefix + ".to_queries", dim, dim)
        fc(prefix + ".to_keys", to_dim, dim)
        fc(prefix + ".to_values", to_dim, dim)
        fc(prefix + ".from_pos_map", pos_dim, dim)
        fc(prefix + ".to_pos_map", pos_dim, dim)
        fc(prefix + ".modulation", dim, dim)

    sd["pos"] = rng.random((T, D)).astype(np.float32)

    for res in cfg.block_resolutions:
        b = f"synthesis.b{res}"
        cout = cfg.channels(res)
        cin = cfg.channels(res // 2) if res > 4 else 0
        att = cfg.has_attention(res)
        if res == 4:
            sd[b + ".const"] = normal(cout, 4, 4)
        sd[b + ".resample_kernel"] = fir_kernel_1331()
        layers = ([("conv0", cin)] if res > 4 else []) + [("conv1", cout)]
        for name, ci in layers:
            p = f"{b}.{name}"
            sd[p + ".weight"] = normal(cout, ci, 3, 3)
            sd[p + ".noise_strength"] = np.float32(noise_strength) * np.ones((), np.float32)
            sd[p + ".resample_kernel"] = fir_kernel_1331()
            sd[p + ".noise_const"] = normal(res, res)
            if att:
                sd[p + ".grid_pos"] = sinusoidal_grid(res, D)
            fc(p + ".affine", D, ci, bias_mean=1.0)
            sd[p + ".biasAct.bias"] = normal(cout, std=0.1)
            if att:
                transformer(p + ".transformer", cout, D, D, kmeans=True)
        if res == cfg.img_resolution:
            p = f"{b}.torgb"
            sd[p + ".weight"] = normal(cfg.img_channels, cout, 1, 1)
            fc(p + ".affine", D, cout, bias_mean=1.0)
            sd[p + ".biasAct.bias"] = normal(cfg.img_channels, std=0.1)
        if res > 4:
            sd[b + ".skip.weight"] = normal(cout, cin, 1, 1)
            sd[b + ".skip.resample_kernel"] = fir_kernel_1331()
        if res == cfg.img_resolution:
            p = f"{b}.conv_last"
            sd[p + ".weight"] = normal(cout, cout, 3, 3)
            sd[p + ".resample_kernel"] = fir_kernel_1331()
            fc(p + ".affine", D, cout, bias_mean=1.0)

    lr = cfg.mapping_lrmul
    n_res = cfg.mapping_layers // 2
    sd["mapping.w_avg"] = np.zeros((D,), np.float32)
    for mlp, with_sa in (("mapping.global_mlp", False), ("mapping.mlp", True)):
        fc(mlp + ".out_layer", D, D, lrmul=lr)
        for i in range(n_res):
            if with_sa:
                transformer(f"{mlp}.sa{i}", D, D, D, kmeans=False)
            fc(f"{mlp}.l{i}.fc0", D, D, lrmul=lr)
            fc(f"{mlp}.l{i}.fc1", D, D, lrmul=lr)
    return sd


def synthetic_latents(cfg: GeneratorConfig, n: int, seed: int) -> np.ndarray:
    """z ~ N(0, I) of shape [n, k, z_dim] (1024_generate.py:19-41 sampling shape)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal((n, cfg.k, cfg.z_dim)).astype(np.float32)
