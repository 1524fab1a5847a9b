"""CPU: libdfhip.so loads and exports every symbol include/dfhip.h declares (no compute calls);
the product refuses to run without a GPU instead of falling back to the CPU."""
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "dfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfhip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from deepflows_amd import _lib
    assert _header_symbols() == sorted(_lib.ALL_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from deepflows_amd import _lib
    assert _lib.library_present(), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    lib = _lib.load()
    missing = [s for s in _header_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_error_reporting_without_gpu_is_loud():
    from deepflows_amd import backend_api, hip_device
    if hip_device.available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError):
        backend_api.Device("hip")
    with pytest.raises(KeyError):
        backend_api.Device("cpu")          # no CPU compute device ships in the product


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "deepflows_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
