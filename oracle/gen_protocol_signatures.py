"""TEST INFRASTRUCTURE -- the device-module protocol as the REFERENCE defines it.  Runs ONLY in the build container
(reference at /root/reference).  Writes tests/golden/protocol_signatures.json:

  * "numpy_backend": for every method of the reference's NumPy device module (the class `NumpyBackend` that
    `cpu_numpy()` builds, DeepFlows/backend/backend_tensor.py:64-172), its parameter names in order
    (`inspect.signature`) -- what BackendTensor calls `device.<method>(...)` with;
  * "cuda_module": the names the reference's CUDA extension exports (`m.def("...")` / `py::class_` in
    DeepFlows/backend/backend_src/ndarray_backend_cuda.cu:515-717), read from the file's text with a regular
    expression -- names only, no source is stored.

tests/test_frontend_host_logic.py asserts the hip device module (deepflows_amd/hip_device.py: HipModule) and the oracle's
NumPy device against this file, so the protocol the new device implements is pinned by the reference, not by this
repository's own mirror (Missing #5).

    python oracle/gen_protocol_signatures.py
"""
import inspect
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "protocol_signatures.json")
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def main():
    from DeepFlows.backend import backend_tensor as ref_bt
    mod = ref_bt.cpu_numpy().mod                       # BackendDevice("cpu", NumpyBackend())
    methods = {}
    for name, fn in inspect.getmembers(type(mod), predicate=inspect.isfunction):
        if name.startswith("__"):
            continue
        params = [p for p in inspect.signature(fn).parameters if p != "self"]
        methods[name] = params
    text = open(os.path.join(REF, "DeepFlows", "backend", "backend_src", "ndarray_backend_cuda.cu")).read()
    body = text[text.index("PYBIND11_MODULE"):]
    cuda_defs = sorted(set(re.findall(r'm\.def\(\s*"([A-Za-z_0-9]+)"', body)))
    cuda_classes = sorted(set(re.findall(r'py::class_<[^>]+>\(\s*m\s*,\s*"([A-Za-z_0-9]+)"', body)))
    cuda_attrs = sorted(set(re.findall(r'm\.attr\(\s*"([A-Za-z_0-9]+)"', body)))
    out = {"source": {"numpy_backend": "DeepFlows/backend/backend_tensor.py:64-172 (inspect.signature of NumpyBackend's methods)",
                      "cuda_module": "DeepFlows/backend/backend_src/ndarray_backend_cuda.cu:515-717 (m.def / py::class_ / m.attr names)"},
           "numpy_backend": methods, "cuda_module": {"functions": cuda_defs, "classes": cuda_classes, "attrs": cuda_attrs}}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT, "-", len(methods), "NumPy-device methods,", len(cuda_defs), "CUDA-module functions")


if __name__ == "__main__":
    main()
