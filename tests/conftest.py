import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


def rel_err(got, want):
    """BASELINE's fp32 parity metric: max|a-b| / max|b| (threshold 1e-4)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"shape {got.shape} vs {want.shape}"
    if want.size == 0:
        return 0.0
    scale = max(float(np.abs(want).max()), 1e-30)
    return float(np.abs(got - want).max()) / scale


def assert_close(got, want, tol=1e-4, what=""):
    e = rel_err(got, want)
    assert e <= tol, f"{what}: relative error {e:.3e} > {tol:g}"


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture()
def cpu_oracle():
    """Registers oracle/numpy_device.py as device 'cpu' (and as the default device) for the
    duration of one test.  The product package never does this by itself."""
    from deepflows_amd import backend_api
    from deepflows_amd.tensor import Graph
    from oracle.numpy_device import NumpyDevice
    mod = NumpyDevice()
    dev = backend_api.register_device("cpu", mod)
    backend_api.set_default_device("cpu")
    Graph.free_graph_all()
    from deepflows_amd.autograd import set_grad_enabled
    set_grad_enabled(True)
    yield dev
    Graph.free_graph_all()
    set_grad_enabled(True)
    backend_api.set_default_device(None)
    backend_api.unregister_device("cpu")


@pytest.fixture()
def hip_device():
    from deepflows_amd import backend_api
    from deepflows_amd.tensor import Graph
    from deepflows_amd.autograd import set_grad_enabled
    dev = backend_api.Device("hip")     # raises loudly when the extension / GPU is missing
    Graph.free_graph_all()
    set_grad_enabled(True)
    dev.mod.conv_dgrad_mode = "overwrite"
    yield dev
    Graph.free_graph_all()
    set_grad_enabled(True)
    dev.mod.conv_dgrad_mode = "overwrite"
