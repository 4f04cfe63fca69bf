import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Soak runs (tools/suite_soak.sh): MGF_SOAK_OFFSET=k shifts every torch seed a test sets -- other random inputs for every HIP-vs-oracle comparison.  Tests that
# re-draw the inputs of a committed fixture by seed are expected to fail under it (they fail on EVERY offset, by a lot); 0 / unset = the committed cases.
_SOAK = int(os.environ.get("MGF_SOAK_OFFSET", "0")) * 1000003
if _SOAK:
    import torch as _torch
    _ms = _torch.manual_seed
    _torch.manual_seed = lambda s: _ms(int(s) + _SOAK)              # (torch.Generator.manual_seed is a C method and stays as it is)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes tens of seconds (full 1024^2 oracle forward)")


def pytest_sessionstart(session):
    """Safety net for a fresh checkout: the library is git-ignored, so build it (hipcc cross-compiles without a GPU) when it is
    missing.  The product itself never does this -- `_lib.lib()` raises if the .so is absent."""
    lib = os.path.join(ROOT, "morphganformer_amd", "libmgf_hip.so")
    if not os.path.exists(lib) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        from morphganformer_amd.build import build
        build()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False, help="also run tests marked slow")
