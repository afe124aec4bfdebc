import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# wide300 / wide2k (round 4): f_in >= 193 -> the default GAE(...) runs layer 1 through gae_xw_fwd / gae_spmm_csr_epilogue /
# gae_xw_wgrad (transform-first order); the vectors are the reference's own act((A H) W^T + b)
CASES = ["tiny", "sym200", "deep3", "single", "mol8", "wide300", "wide2k"]
WIDE_CASES = ["wide300", "wide2k"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """where two GPUs are visible the 2-rank RCCL test runs FIRST: it is the only evidence the N > 1 path has on
    hardware, so it must not sit behind a failure or a time-out of anything else"""
    first = [i for i in items if i.name == "test_two_rank_rccl_sharded_encoder"]
    if first:
        items[:] = first + [i for i in items if i not in first]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_params(g):
    """(weights, biases) lists in layer order from the reference state-dict keys."""
    L = len(g["hidden"])
    Ws = [g[f"sd/layers.{i}.apply_mod.linear.weight"] for i in range(L)]
    bs = [g[f"sd/layers.{i}.apply_mod.linear.bias"] for i in range(L)]
    return Ws, bs


@pytest.fixture(params=CASES)
def golden(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g


@pytest.fixture
def tuning():
    """set library knobs (gae_tuning_set) for one test; the previous values come back afterwards"""
    from gae_dgl_amd import _lib
    saved = []

    def set_knob(name, value):
        import ctypes
        old = ctypes.c_int64(0)
        _lib.call("gae_tuning_get", name.encode(), ctypes.byref(old))
        saved.append((name, old.value))
        _lib.call("gae_tuning_set", name.encode(), int(value))
    yield set_knob
    for name, value in reversed(saved):
        _lib.call("gae_tuning_set", name.encode(), value)
