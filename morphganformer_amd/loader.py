"""Checkpoint reader for the reference's network pickles -- `load_network(path) -> {"G", "D", "Gs"}` like loader.py:26-47 --
WITHOUT executing anything stored in the file.

Layout (training/training_loop.py:113-132, torch_utils/persistence.py:110-118,171-194): the snapshot is a
pickled dict `{G, D, Gs, ...}`; every `@persistent_class` instance reduces to
`torch_utils.persistence._reconstruct_persistent_obj(meta)` with `meta = {type: 'class', version, module_src, class_name, state}`
where `state` is the module's `__dict__` (`_parameters`, `_buffers`, `_modules`, `_init_args`, `_init_kwargs`, plain attributes).
The reference rebuilds objects by `exec`-ing `module_src`; this reader instead intercepts the reconstruct call, ignores
`module_src`, and keeps `(class_name, state)` as an inert stub tree that is flattened into a `state_dict`.  The unpickler only
resolves an allow-list of globals (torch tensor rebuild helpers, numpy array rebuild helpers, OrderedDict, EasyDict); anything
else raises.

Legacy TensorFlow snapshots (loader.py:36-41: a 3-tuple of `dnnlib.tflib.network.Network` objects) are read the same inert way
(`TFNetworkStub`) and their generator variables renamed / transposed / flipped into the PyTorch state_dict layout by
`convert_tf_generator` -- the rules of loader.py:91-247 restated as a parser of the TensorFlow variable names.
"""
from __future__ import annotations

import collections
import io
import math
import pickle
import re
from dataclasses import dataclass

import numpy as np
import torch

from .synth_weights import GeneratorConfig


class EasyDict(dict):
    """Attribute-access dict (dnnlib/util.py:32-44)."""
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


@dataclass
class PersistentStub:
    """Inert stand-in for one pickled `@persistent_class` instance."""
    class_name: str
    state: dict
    version: int = 0

    @property
    def init_kwargs(self):
        return dict(self.state.get("_init_kwargs", {}))

    def state_dict(self, prefix=""):
        out = collections.OrderedDict()
        for name, p in (self.state.get("_parameters") or {}).items():
            if p is not None:
                out[prefix + name] = p.detach() if isinstance(p, torch.Tensor) else torch.as_tensor(p)
        for name, b in (self.state.get("_buffers") or {}).items():
            if b is not None and name not in (self.state.get("_non_persistent_buffers_set") or ()):
                out[prefix + name] = b
        for name, m in (self.state.get("_modules") or {}).items():
            if isinstance(m, (PersistentStub, InertModule)):
                out.update(m.state_dict(prefix + name + "."))
            elif m is not None:
                raise pickle.UnpicklingError(f"sub-module {prefix + name} is not a persistent object ({type(m).__name__})")
        return out


class InertModule:
    """Stand-in for a plain `torch.nn.modules.*` child (e.g. the attention Dropout, networks.py:592): the unpickler creates
    it with NEWOBJ and hands the module `__dict__` to `__setstate__`; no torch.nn code runs."""
    class_name = "Module"

    def __init__(self, *a, **k):
        self.state = {}

    def __setstate__(self, state):
        self.state = dict(state)

    state_dict = PersistentStub.state_dict
    init_kwargs = PersistentStub.init_kwargs


def _inert_class(name):
    return type(name, (InertModule,), {"class_name": name})


class TFNetworkStub(EasyDict):
    """Inert stand-in for a pickled `dnnlib.tflib.network.Network` (the reference maps it onto an EasyDict too, loader.py:51-58):
    `__setstate__` only stores version / static_kwargs / variables / components; the build-function source is ignored."""

    def __setstate__(self, state):
        self.update(state)


def _reconstruct_persistent_obj(meta):
    meta = dict(meta)
    if meta.get("type") != "class":
        raise pickle.UnpicklingError(f"unsupported persistent object type {meta.get('type')!r}")
    state = meta.get("state")
    return PersistentStub(str(meta["class_name"]), dict(state) if state is not None else {}, int(meta.get("version", 0)))


_ALLOWED = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("dnnlib.util", "EasyDict"): EasyDict, (__name__, "EasyDict"): EasyDict,
    ("dnnlib.tflib.network", "Network"): TFNetworkStub, (__name__, "TFNetworkStub"): TFNetworkStub,
    ("torch_utils.persistence", "_reconstruct_persistent_obj"): _reconstruct_persistent_obj,
    ("builtins", "set"): set, ("builtins", "dict"): dict, ("builtins", "list"): list, ("builtins", "tuple"): tuple,
    ("builtins", "slice"): slice, ("builtins", "complex"): complex, ("builtins", "frozenset"): frozenset,
}
_ALLOWED_PREFIXES = {
    "torch._utils": {"_rebuild_tensor_v2", "_rebuild_parameter", "_rebuild_tensor", "_rebuild_parameter_with_state"},
    "torch": {"FloatStorage", "DoubleStorage", "HalfStorage", "LongStorage", "IntStorage", "ShortStorage", "CharStorage",
              "ByteStorage", "BoolStorage", "BFloat16Storage", "Size", "device", "float32", "float64", "float16", "int64", "int32"},
    "torch.storage": {"_load_from_bytes", "UntypedStorage", "TypedStorage"},
    "torch.nn.parameter": {"Parameter"},
    "numpy.core.multiarray": {"_reconstruct", "scalar"}, "numpy._core.multiarray": {"_reconstruct", "scalar"},
    "numpy": {"ndarray", "dtype"},
    "numpy.core.numeric": {"_frombuffer"}, "numpy._core.numeric": {"_frombuffer"},
}


class SafeNetworkUnpickler(pickle.Unpickler):
    """Resolves only the globals a network snapshot legitimately needs; never imports or executes embedded source."""

    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return _ALLOWED[(module, name)]
        if module.startswith("torch.nn.modules."):
            return _inert_class(name)
        if name in _ALLOWED_PREFIXES.get(module, ()):
            mod = __import__(module, fromlist=[name])
            return getattr(mod, name)
        raise pickle.UnpicklingError(f"refusing to resolve global {module}.{name} from a network pickle")


def _load_bytes_safely(b):
    # torch tensors pickled outside torch.save carry their storage through torch.storage._load_from_bytes -> torch.load;
    # force weights_only semantics there
    return torch.load(io.BytesIO(b), weights_only=True, map_location="cpu")


def read_pickle(path_or_file):
    f = open(path_or_file, "rb") if isinstance(path_or_file, (str, bytes)) or hasattr(path_or_file, "__fspath__") else path_or_file
    try:
        import torch.storage as ts
        orig = ts._load_from_bytes
        ts._load_from_bytes = _load_bytes_safely
        try:
            return SafeNetworkUnpickler(f).load()
        finally:
            ts._load_from_bytes = orig
    finally:
        if f is not path_or_file:
            f.close()


def config_from_stub(stub: PersistentStub) -> GeneratorConfig:
    """GeneratorConfig from the pickled Generator's constructor arguments (training/networks.py:1269-1302)."""
    kw = stub.init_kwargs
    syn = dict(kw.get("synthesis_kwargs", {}))
    mp = dict(kw.get("mapping_kwargs", {}))
    return GeneratorConfig(
        img_resolution=int(kw["img_resolution"]), img_channels=int(kw.get("img_channels", 3)), z_dim=int(kw["z_dim"]),
        w_dim=int(kw["w_dim"]), k=int(kw["k"]), channel_base=int(syn.get("channel_base", 32 << 10)),
        channel_max=int(syn.get("channel_max", 512)), attn_max_log2res=int(syn.get("end_res", 20)),
        mapping_layers=int(mp.get("num_layers", 8)), mapping_lrmul=float(mp.get("lrmul", 0.01)),
        normalize_global=bool(mp.get("normalize_global", True)))


# ------------------------------------------------------------------------------------------------------ legacy TensorFlow pickles
def collect_tf_params(tf_net, prefix=""):
    """Flatten `variables` of a network and of its components into {"comp/sub/name": ndarray} (loader.py:60-68)."""
    out = {}
    for name, value in tf_net["variables"]:
        out[prefix + name] = np.asarray(value)
    for name, comp in tf_net.get("components", {}).items():
        out.update(collect_tf_params(comp, prefix + name + "/"))
    return out


def config_from_tf_kwargs(kw) -> GeneratorConfig:
    """GeneratorConfig from a TF network's static_kwargs (the kwarg translation of loader.py:97-154).  Only the architecture this
    engine implements is accepted: resnet blocks, duplex (k-means) attention with sinusoidal positions, multiplicative
    integration -- what the published GANformer snapshots use."""
    get = lambda k, d: d if kw.get(k, d) is None else kw.get(k, d)
    need = {"architecture": "resnet", "transformer": True, "style": True, "local_noise": True, "kmeans": True, "use_pos": True,
            "integration": "mul", "norm": "layer", "mapping_resnet": True, "mapping_ltnt2ltnt": True}
    for k, v in need.items():
        if kw.get(k) != v:
            raise NotImplementedError(f"TensorFlow snapshot built with {k}={kw.get(k)!r}; this engine implements {k}={v!r}")
    if get("label_size", 0) != 0 or get("num_heads", 1) != 1 or get("start_res", 0) != 0:
        raise NotImplementedError("conditional / multi-head / start_res > 0 snapshots are not supported")
    return GeneratorConfig(
        img_resolution=int(get("resolution", 1024)), img_channels=int(get("num_channels", 3)), z_dim=int(get("latent_size", 512)),
        w_dim=int(get("dlatent_size", 512)), k=int(get("components_num", 1)) + 1, channel_base=int(get("fmap_base", 16 << 10)) * 2,
        channel_max=int(get("fmap_max", 512)), attn_max_log2res=int(get("end_res", 8)),
        mapping_layers=int(get("mapping_layersnum", 8)), mapping_lrmul=float(get("mapping_lrmul", 0.01)), normalize_global=False)


_QKV = {"query": "to_queries", "key": "to_keys", "value": "to_values"}


def _att_entry(rest):
    """'weight_query' / 'bias_from_pos' / 'weight_out' / 'toasgn_init' / 'iter_0/st_weights' -> (torch suffix, transpose?)"""
    if rest == "toasgn_init":
        return "centroids", False
    if rest == "iter_0/st_weights":
        return "att_weight", False
    kind, _, what = rest.partition("_")
    if kind not in ("weight", "bias"):
        return None, False
    if what in _QKV:
        mod = _QKV[what]
    elif what in ("from_pos", "to_pos"):
        mod = what + "_map"
    elif what == "out":
        mod = "modulation"
    else:
        return None, False          # key2 (queries2centroids) and friends: not part of the PyTorch module
    return f"{mod}.{kind}", kind == "weight"


def convert_tf_generator(tf_G):
    """TF generator stub -> (state_dict of numpy float32 arrays under the PyTorch key names, GeneratorConfig).
    Transforms (loader.py:185-240): dense weights are transposed; conv kernels go HWIO -> OIHW, and the up-sampling conv0 and the
    skip kernels are additionally flipped in H and W; every style affine bias gets +1; `dlatent_avg` -> mapping.w_avg,
    `ltnt_emb/emb` -> pos, `synthesis/noise<j>` -> the j-th layer's noise_const; position grids and FIR kernels are rebuilt."""
    if int(tf_G.get("version", 0)) < 4:
        raise ValueError("TensorFlow pickle version too low")
    cfg = config_from_tf_kwargs(dict(tf_G["static_kwargs"]))
    tfp = collect_tf_params(tf_G)
    sd = {}
    f32 = lambda a: np.array(a, dtype=np.float32, order="C")          # (ascontiguousarray would turn 0-d into 1-d)
    res_out = cfg.img_resolution
    for name, v in tfp.items():
        if name == "dlatent_avg":
            sd["mapping.w_avg"] = f32(v)
        elif name == "ltnt_emb/emb":
            sd["pos"] = f32(v)
        elif name.startswith("mapping/"):
            rest = name[len("mapping/"):]
            mlp = "mlp"
            if rest.startswith("global/"):
                mlp, rest = "global_mlp", rest[len("global/"):]
            layer, _, leaf = rest.partition("/")
            m = re.fullmatch(r"Dense(\d+)_(\d+)", layer)
            if m and leaf in ("weight", "bias"):
                sd[f"mapping.{mlp}.l{m.group(1)}.fc{m.group(2)}.{leaf}"] = f32(v.T if leaf == "weight" else v)
            elif layer == "Dense3" and leaf in ("weight", "bias"):                # the output layer's fixed name (loader.py:195-196)
                sd[f"mapping.{mlp}.out_layer.{leaf}"] = f32(v.T if leaf == "weight" else v)
            elif layer.startswith("AttLayer_") and mlp == "mlp":
                suffix, tr = _att_entry(leaf)
                if suffix is not None and not suffix.startswith(("centroids", "att_weight")):
                    sd[f"mapping.mlp.sa{layer[len('AttLayer_'):]}.{suffix}"] = f32(v.T if tr else v)
        elif name.startswith("synthesis/"):
            rest = name[len("synthesis/"):]
            m = re.fullmatch(r"noise(\d+)", rest)
            if m:
                j = int(m.group(1))                                               # layer j: res 2^((j+5)//2), conv (j+5) % 2 ... inverse of
                log2r, i = divmod(j + 5, 2)                                       # j = log2(r)*2 - 5 + i  (loader.py:222)
                sd[f"synthesis.b{2 ** log2r}.conv{i}.noise_const"] = f32(v[0, 0])
                continue
            m = re.fullmatch(r"(\d+)x\1/(.+)", rest)
            if not m:
                continue
            r, leaf = int(m.group(1)), m.group(2)
            b = f"synthesis.b{r}"
            if leaf == "Const/const":
                sd[b + ".const"] = f32(v[0])
                continue
            layer, _, what = leaf.partition("/")
            if layer in ("Conv", "Conv1", "Conv0_up"):
                conv = "conv0" if layer == "Conv0_up" else "conv1"
                if what == "weight":
                    w = v[::-1, ::-1] if conv == "conv0" else v                   # the transposed conv stores the flipped kernel
                    sd[f"{b}.{conv}.weight"] = f32(w.transpose(3, 2, 0, 1))
                elif what == "bias":
                    sd[f"{b}.{conv}.biasAct.bias"] = f32(v)
                elif what == "noise_strength":
                    sd[f"{b}.{conv}.noise_strength"] = f32(v)
                elif what == "mod_weight":
                    sd[f"{b}.{conv}.affine.weight"] = f32(v.T)
                elif what == "mod_bias":
                    sd[f"{b}.{conv}.affine.bias"] = f32(v + 1)
                elif what.startswith("AttLayer_l2n/"):
                    suffix, tr = _att_entry(what[len("AttLayer_l2n/"):])
                    if suffix is not None:
                        sd[f"{b}.{conv}.transformer.{suffix}"] = f32(v.T if tr else v)
            elif layer == "Skip" and what == "weight":
                sd[b + ".skip.weight"] = f32(v[::-1, ::-1].transpose(3, 2, 0, 1))
            elif layer == "ToRGB" and r == res_out:
                if what == "weight":
                    sd[b + ".torgb.weight"] = f32(v.transpose(3, 2, 0, 1))
                elif what == "bias":
                    sd[b + ".torgb.biasAct.bias"] = f32(v)
                elif what == "mod_weight":
                    sd[b + ".torgb.affine.weight"] = f32(v.T)
                elif what == "mod_bias":
                    sd[b + ".torgb.affine.bias"] = f32(v + 1)
                elif what == "extraLayer/weight":
                    sd[b + ".conv_last.weight"] = f32(v.transpose(3, 2, 0, 1))
                elif what == "extraLayer/mod_weight":
                    sd[b + ".conv_last.affine.weight"] = f32(v.T)
                elif what == "extraLayer/mod_bias":
                    sd[b + ".conv_last.affine.bias"] = f32(v + 1)
    # buffers the TF snapshot does not carry: FIR kernels and sinusoidal position grids are functions of the config
    from .synth_weights import make_state_dict
    template = make_state_dict(cfg, seed=0)
    for k, v in template.items():
        if k.endswith("resample_kernel") or k.endswith("grid_pos"):
            sd[k] = v
    missing = [k for k in template if k not in sd]
    if missing:
        raise pickle.UnpicklingError(f"TensorFlow snapshot lacks {len(missing)} generator variables, e.g. {missing[:4]}")
    for k, v in template.items():
        if tuple(np.shape(sd[k])) != tuple(np.shape(v)):
            raise pickle.UnpicklingError(f"{k}: TensorFlow variable has shape {np.shape(sd[k])}, expected {np.shape(v)}")
    return {k: sd[k] for k in template}, cfg


def load_network_stubs(path):
    data = read_pickle(path)
    if isinstance(data, tuple):
        if not (len(data) == 3 and all(isinstance(net, TFNetworkStub) for net in data)):
            raise pickle.UnpicklingError("a tuple snapshot must hold the three TensorFlow networks (G, D, Gs)")
        return dict(zip(("G", "D", "Gs"), data))
    for key in ("G", "D", "Gs"):
        if key not in data or not isinstance(data[key], PersistentStub):
            raise pickle.UnpicklingError(f"snapshot has no persistent network under key {key!r}")
    return data


def load_network(path, device="cuda", which=("Gs",)):
    """{"G", "D", "Gs"} like the reference; generators listed in `which` become HIP `engine.Generator` objects, the other
    entries stay inert stubs (the discriminator is never used by the projection path)."""
    from .engine import Generator
    data = load_network_stubs(path)
    out = dict(data)
    for key in which:
        stub = data[key]
        if isinstance(stub, TFNetworkStub):
            sd, cfg = convert_tf_generator(stub)
            out[key] = Generator(sd, cfg, device)
            continue
        if stub.class_name != "Generator":
            raise pickle.UnpicklingError(f"{key} is a {stub.class_name}, expected Generator")
        sd = {k: v.detach().cpu().numpy() for k, v in stub.state_dict().items()}
        out[key] = Generator(sd, config_from_stub(stub), device)
    return out


def save_snapshot_like_reference(path, networks: dict, class_names: dict, init_kwargs: dict):
    """Test helper: write `{key: state_dict}` in the reference's persistent-pickle layout (empty `module_src`)."""
    import sys
    import types

    class _Obj:
        def __init__(self, meta):
            self.meta = meta

        def __reduce__(self):
            return (_reconstruct_persistent_obj, (self.meta,))

    def tree(class_name, sd, kwargs):
        root = {"_parameters": collections.OrderedDict(), "_buffers": collections.OrderedDict(), "_modules": collections.OrderedDict(),
                "_init_args": (), "_init_kwargs": kwargs, "training": False}
        children = collections.OrderedDict()
        for k, v in sd.items():
            head, _, rest = k.partition(".")
            if rest:
                children.setdefault(head, collections.OrderedDict())[rest] = v
            else:
                root["_buffers"][head] = torch.as_tensor(v)
        for name, sub in children.items():
            root["_modules"][name] = tree("Module", sub, {})
        return _Obj(EasyDict(type="class", version=6, module_src="", class_name=class_name, state=root))

    # make the reduce target pickle under the reference's module path
    mod = types.ModuleType("torch_utils.persistence")
    mod._reconstruct_persistent_obj = _reconstruct_persistent_obj
    pkg = types.ModuleType("torch_utils")
    saved = {k: sys.modules.get(k) for k in ("torch_utils", "torch_utils.persistence")}
    sys.modules["torch_utils"], sys.modules["torch_utils.persistence"] = pkg, mod
    old_mod, old_qual = _reconstruct_persistent_obj.__module__, _reconstruct_persistent_obj.__qualname__
    _reconstruct_persistent_obj.__module__ = "torch_utils.persistence"
    try:
        payload = {k: tree(class_names[k], sd, init_kwargs.get(k, {})) for k, sd in networks.items()}
        with open(path, "wb") as f:
            pickle.dump(payload, f)
    finally:
        _reconstruct_persistent_obj.__module__ = old_mod
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
