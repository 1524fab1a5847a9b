"""Checkpoints in the reference's format (DeepFlows/utils/model_utils.py:19-181).

The file is a pickled dict of plain Python values and NumPy arrays:

    {'epoch': int, 'loss': float | None,
     'model_parameters': {named_parameters() name: float32 ndarray},
     'optimizer_state': {'type': class name, 'lr' / 'momentum' / 'weight_decay' if present,
                         'v': [ndarray per optimizer parameter], 's': [...], 't': int}}   # Adam

so a checkpoint written by the reference (e.g. test/checkpoints-cifar10cuda_70%/*.pkl) loads into
the `hip` models and the other way round.  Two deliberate differences from the reference loader:
  * unpickling is restricted to NumPy arrays / dtypes / scalars and builtins (the reference calls
    dill/pickle.load on whatever file it is given);
  * a parameter whose stored shape does not match raises instead of being silently replaced.
BatchNorm running statistics are plain attributes in the reference and are NOT part of its
checkpoints (batchnorm.py:23-28); an optional 'buffers' entry carries them here and is ignored by
the reference's loader.
"""
import io
import os
import pickle

import numpy as np

from .. import backend_api
from ..backend_selection import BackendTensor

_ALLOWED = {
    ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy", "float32"), ("numpy", "float64"), ("numpy", "int64"),
    ("numpy", "int32"), ("numpy", "bool_"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "float"),
    ("builtins", "int"), ("builtins", "str"), ("builtins", "bool"), ("builtins", "complex"), ("builtins", "bytes"),
    ("builtins", "bytearray"), ("collections", "OrderedDict"),
    ("dill._dill", "_load_type"),       # dill writes `dict`/`list` type references through this helper
    ("dill._dill", "_create_array"),    # ... and every ndarray through this one (the reference saves with dill when it is installed)
}


def _dill_create_array(f, args, state, npdict=None):
    """What dill._dill._create_array does for a plain ndarray, without importing dill: `f` must be NumPy's
    _reconstruct (it came through find_class, so it is), the array is rebuilt from (args, state); an attached
    __dict__ (ndarray subclasses) is refused."""
    if getattr(f, "__name__", "") != "_reconstruct":
        raise pickle.UnpicklingError("checkpoint: dill array record does not use numpy's _reconstruct")
    if npdict:
        raise pickle.UnpicklingError("checkpoint: ndarray subclasses with attributes are not loaded")
    array = f(*args)
    array.__setstate__(state)
    return array


_DILL_TYPES = {"dict": dict, "list": list, "tuple": tuple, "float": float, "int": int, "str": str, "bool": bool,
               "NoneType": type(None)}


class _ArrayUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            if module == "dill._dill":      # shims: dill itself is never imported by the loader
                return _dill_create_array if name == "_create_array" else (lambda type_name: _DILL_TYPES[type_name])
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}: only NumPy arrays and builtins are loaded")


def _to_numpy(t):
    t = t.data if hasattr(t, "data") and not isinstance(t, BackendTensor) else t
    return np.ascontiguousarray(t.numpy(), dtype=np.float32)


def _optimizer_params(optimizer):
    return getattr(optimizer, "params", None) or getattr(optimizer, "parameters", [])


def save_checkpoint(model, optimizer=None, epoch=0, loss=None, save_path="checkpoint.pkl", buffers=True):
    """Same signature and file layout as the reference's save_checkpoint (:19-83)."""
    os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
    ck = {"epoch": epoch, "loss": loss, "model_parameters": {}}
    for name, p in model.named_parameters():
        ck["model_parameters"][name] = _to_numpy(p)
    if buffers:
        bufs = {}
        for mname, m in model.named_modules():
            for attr in ("running_mean", "running_var"):
                t = getattr(m, attr, None) if attr in getattr(m, "__dict__", {}) or hasattr(m, attr) else None
                if t is not None and hasattr(t, "numpy"):
                    bufs[(mname + "." if mname else "") + attr] = _to_numpy(t)
        if bufs:
            ck["buffers"] = bufs
    if optimizer is not None:
        st = {"type": type(optimizer).__name__}
        for attr in ("lr", "momentum", "weight_decay"):
            if hasattr(optimizer, attr):
                st[attr] = getattr(optimizer, attr)
        if hasattr(optimizer, "v"):
            st["v"] = [_to_numpy(v) for v in optimizer.v]
        if hasattr(optimizer, "s"):
            st["s"] = [_to_numpy(s) for s in optimizer.s]
        if hasattr(optimizer, "t"):
            st["t"] = int(optimizer.t)
        ck["optimizer_state"] = st
    with open(save_path, "wb") as f:
        pickle.dump(ck, f, protocol=4)


def read_checkpoint(save_path):
    """The checkpoint dict, unpickled with the restricted loader."""
    if not os.path.exists(save_path):
        raise FileNotFoundError(f"checkpoint file does not exist: {save_path}")
    with open(save_path, "rb") as f:
        data = f.read()
    ck = _ArrayUnpickler(io.BytesIO(data)).load()
    if not isinstance(ck, dict) or "model_parameters" not in ck:
        raise RuntimeError("not a DeepFlows checkpoint (no 'model_parameters')")
    return ck


def _device_array(arr, like):
    dev = like.device
    return backend_api.Btensor(np.ascontiguousarray(arr, dtype=np.float32), dtype="float32", device=dev) \
        if hasattr(backend_api, "Btensor") else BackendTensor(np.ascontiguousarray(arr, dtype=np.float32), device=dev)


def _assign(holder, arr, like=None):
    """Put `arr` into the tensor `holder` (a Tensor: its .data is replaced or overwritten) IN PLACE when the existing
    device array is a compact buffer of the same size -- a step captured in a hipGraph (graph_step.CapturedTrainStep)
    has the raw pointers of parameters, Adam moments and running statistics baked in, so a load must not move them."""
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    cur = holder.data if hasattr(holder, "data") and not isinstance(holder, BackendTensor) else holder
    handle = getattr(cur, "_handle", None)
    if (handle is not None and tuple(int(s) for s in cur.shape) == tuple(arr.shape) and
            getattr(cur, "_offset", 0) == 0 and getattr(handle, "size", -1) == arr.size and cur.is_compact()):
        cur.device.from_numpy(arr, handle)
        return cur
    return _device_array(arr, like if like is not None else holder)


def load_checkpoint(model, optimizer=None, save_path="checkpoint.pkl"):
    """Same signature and return value as the reference's load_checkpoint (:86-181)."""
    ck = read_checkpoint(save_path)
    stored = ck["model_parameters"]
    for name, p in model.named_parameters():
        if name in stored:
            a = np.asarray(stored[name], dtype=np.float32)
            if tuple(a.shape) != tuple(int(s) for s in p.shape):
                raise ValueError(f"checkpoint parameter {name} has shape {a.shape}, model has {tuple(p.shape)}")
            p.data = _assign(p, a)
    bufs = ck.get("buffers") or {}
    if bufs:
        mods = dict(model.named_modules())
        for key, a in bufs.items():
            mname, _, attr = key.rpartition(".")
            m = mods.get(mname)
            t = getattr(m, attr, None) if m is not None else None
            if t is not None:
                t.data = _assign(t, a)
    if optimizer is not None and "optimizer_state" in ck:
        st = ck["optimizer_state"]
        for attr in ("lr", "momentum", "weight_decay"):
            if attr in st and hasattr(optimizer, attr):
                setattr(optimizer, attr, st[attr])
        params = _optimizer_params(optimizer)
        for key in ("v", "s"):
            if hasattr(optimizer, key) and key in st:
                cur = getattr(optimizer, key)
                for i, p in enumerate(params):
                    if i < len(st[key]) and isinstance(st[key][i], np.ndarray):
                        cur[i] = _assign(cur[i], st[key][i], like=p) if cur[i] is not None else _device_array(st[key][i], p)
        if hasattr(optimizer, "t") and "t" in st:
            optimizer.t = st["t"]
    return {"epoch": ck.get("epoch", 0), "loss": ck.get("loss")}
