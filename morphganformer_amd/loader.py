"""Checkpoint reader for the reference's network pickles -- `load_network(path) -> {"G", "D", "Gs"}` like loader.py:26-47 --
WITHOUT executing anything stored in the file.

Layout (training/training_loop.py:113-132, torch_utils/persistence.py:110-118,171-194): the snapshot is a
pickled dict `{G, D, Gs, ...}`; every `@persistent_class` instance reduces to
`torch_utils.persistence._reconstruct_persistent_obj(meta)` with `meta = {type: 'class', version, module_src, class_name, state}`
where `state` is the module's `__dict__` (`_parameters`, `_buffers`, `_modules`, `_init_args`, `_init_kwargs`, plain attributes).
The reference rebuilds objects by `exec`-ing `module_src`; this reader instead intercepts the reconstruct call, ignores
`module_src`, and keeps `(class_name, state)` as an inert stub tree that is flattened into a `state_dict`.  The unpickler only
resolves an allow-list of globals (torch tensor rebuild helpers, numpy array rebuild helpers, OrderedDict, EasyDict); anything
else raises.  Legacy TensorFlow pickles (loader.py:36-41,91-247) are not converted yet.
"""
from __future__ import annotations

import collections
import io
import pickle
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


def _reconstruct_persistent_obj(meta):
    meta = dict(meta)
    if meta.get("type") != "class":
        raise pickle.UnpicklingError(f"unsupported persistent object type {meta.get('type')!r}")
    state = meta.get("state")
    return PersistentStub(str(meta["class_name"]), dict(state) if state is not None else {}, int(meta.get("version", 0)))


_ALLOWED = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("dnnlib.util", "EasyDict"): EasyDict, (__name__, "EasyDict"): EasyDict,
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


def load_network_stubs(path):
    data = read_pickle(path)
    if isinstance(data, tuple):
        raise NotImplementedError("legacy TensorFlow network pickles (loader.py:36-41) are not supported yet")
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
