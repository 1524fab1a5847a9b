"""Strided ND-array front-end over a device module — the host-side mirror of the reference's
protocol owner, DeepFlows/backend/backend_tensor.py (BackendDevice :11-51, device registry
:54-196, BackendTensor :199-689, module helpers :692-779).

Same names, argument meaning and view semantics (views never copy; `compact()` materialises;
binary ops broadcast only `other`; reductions are single-axis; `mean` divides by the TOTAL
element count, :659-662).  What differs is the registry: the product ships exactly one compute
device, `hip` (libdfhip.so on an MI355X).  There is no CPU compute path in this package; tests
and the CPU-baseline leg of bench.py register the NumPy oracle device explicitly through
`register_device` — nothing falls back to it.
"""
import math
from functools import reduce as _reduce

import numpy as np


def prod(xs):
    out = 1
    for v in xs:
        out *= v
    return out


class BackendDevice:
    """Named wrapper that forwards every attribute to the device module
    (backend_tensor.py:11-28)."""

    def __init__(self, name, mod):
        self.name = name
        self.mod = mod

    def __eq__(self, other):
        return isinstance(other, BackendDevice) and self.name == other.name

    def __hash__(self):
        return hash(self.name)

    def __repr__(self):
        return self.name + "()"

    def __getattr__(self, attr):
        mod = self.__dict__.get("mod")
        if mod is None:
            raise AttributeError(f"device '{self.__dict__.get('name')}' is not available")
        return getattr(mod, attr)

    def enabled(self):
        return self.mod is not None

    def randn(self, *shape, dtype="float32"):
        return BackendTensor(np.random.randn(*shape).astype(dtype), device=self)

    def rand(self, *shape, dtype="float32"):
        return BackendTensor(np.random.rand(*shape).astype(dtype), device=self)

    def one_hot(self, n, i, dtype="float32"):
        return BackendTensor(np.eye(n, dtype=dtype)[i], device=self)

    def empty(self, shape, dtype="float32"):
        assert dtype in (None, "float32")
        return BackendTensor.make(shape, device=self)

    def full(self, shape, fill_value, dtype="float32"):
        assert dtype in (None, "float32")
        out = BackendTensor.make(shape, device=self)
        out.fill(fill_value)
        return out


# ---- device registry -----------------------------------------------------------------------
_hip_device = None
_extra = {}          # name -> BackendDevice registered by tests / the CPU-baseline harness
_default_name = None


def hip():
    """The MI355X device (try-import pattern of the reference's cuda(), :54-61): a cached
    singleton; `.enabled()` is False when libdfhip.so is not built or no GPU is visible."""
    global _hip_device
    if _hip_device is None or not _hip_device.enabled():
        from .. import hip_device
        mod = hip_device.get_module() if hip_device.available() else None
        _hip_device = BackendDevice("hip", mod)
    return _hip_device


def cuda():
    """Scripts written for the reference say device='cuda'; on this backend that is the hip
    device (there is no CUDA here)."""
    return hip()


def register_device(name, mod):
    """Plug a device module in beside `hip` (used by tests and bench.py's cpu_baseline leg to
    register oracle/numpy_device.py as 'cpu'). Returns the BackendDevice."""
    dev = BackendDevice(name, mod)
    _extra[name] = dev
    return dev


def unregister_device(name):
    _extra.pop(name, None)


def set_default_device(name):
    global _default_name
    _default_name = name


def all_devices():
    devs = {"hip": hip(), "cuda": hip()}
    devs.update(_extra)
    return devs


def Device(device_name=None):
    if isinstance(device_name, BackendDevice):
        return device_name
    if device_name is None:
        return default_device()
    if device_name in _extra:
        return _extra[device_name]
    if device_name in ("hip", "cuda"):
        dev = hip()
        if not dev.enabled():
            from .. import hip_device
            raise RuntimeError(
                "device 'hip' is not available: " + (hip_device.unavailable_reason or "libdfhip.so is not built or no "
                "MI355X is visible") + " (there is no CPU fallback in deepflows_amd)")
        return dev
    raise KeyError(f"unknown device '{device_name}' (known: {sorted(all_devices())})")


def cpu():
    return Device("cpu")


def default_device():
    return Device(_default_name if _default_name is not None else "hip")


# ---- the strided array -----------------------------------------------------------------------
class BackendTensor:
    def __init__(self, other, device=None):
        if isinstance(other, BackendTensor):
            device = other.device if device is None else device
            self._adopt(other.to(device) + 0.0)     # copy = one scalar_add (:206-209)
        elif isinstance(other, np.ndarray):
            device = device if device is not None else default_device()
            arr = BackendTensor.make(other.shape, device=device)
            with np.errstate(over="ignore", invalid="ignore"):     # tensor.empty() hands over garbage
                host = np.ascontiguousarray(other.astype(np.float32))
            arr.device.from_numpy(host, arr._handle)
            self._adopt(arr)
        else:
            self._adopt(BackendTensor(np.array(other), device))

    def _adopt(self, src):
        self._shape, self._strides, self._offset = src._shape, src._strides, src._offset
        self._device, self._handle = src._device, src._handle

    @staticmethod
    def compact_strides(shape):
        strides, acc = [], 1
        for extent in reversed(tuple(shape)):
            strides.append(acc)
            acc *= extent
        return tuple(reversed(strides))

    @staticmethod
    def make(shape, strides=None, device=None, handle=None, offset=0):
        t = BackendTensor.__new__(BackendTensor)
        t._shape = tuple(shape)
        t._strides = BackendTensor.compact_strides(shape) if strides is None else strides
        t._offset = offset
        t._device = device if device is not None else default_device()
        t._handle = t._device.Array(prod(shape)) if handle is None else handle
        return t

    shape = property(lambda self: self._shape)
    strides = property(lambda self: self._strides)
    device = property(lambda self: self._device)
    dtype = property(lambda self: "float32")
    ndim = property(lambda self: len(self._shape))
    size = property(lambda self: prod(self._shape))

    def __repr__(self):
        return f"BackendTensor({self.numpy()}, device={self.device})"

    def __str__(self):
        return str(self.numpy())

    def __len__(self):
        return self._shape[0]

    def fill(self, value):
        self._device.fill(self._handle, value)

    def to(self, device):
        return self if device == self.device else BackendTensor(self.numpy(), device=device)

    def numpy(self):
        return self.device.to_numpy(self._handle, self.shape, self.strides, self._offset)

    def is_compact(self):
        return (tuple(self._strides) == self.compact_strides(self._shape)
                and prod(self._shape) == self._handle.size)

    def compact(self):
        if self.is_compact():
            return self
        out = BackendTensor.make(self.shape, device=self.device)
        self.device.compact(self._handle, out._handle, self.shape, self.strides, self._offset)
        return out

    def as_strided(self, shape, strides):
        assert len(shape) == len(strides)
        return BackendTensor.make(shape, strides=strides, device=self.device, handle=self._handle)

    @property
    def flat(self):
        return self.reshape((self.size,))

    def reshape(self, new_shape):
        new_shape = tuple(new_shape)
        # -1 is resolved in the first or last position only (:351-364)
        if new_shape and new_shape[0] == -1:
            rest = prod(new_shape[1:])
            new_shape = (int(self.size / rest),) + new_shape[1:]
        elif new_shape and new_shape[-1] == -1:
            rest = prod(new_shape[:-1])
            new_shape = new_shape[:-1] + (int(self.size / rest),)
        if prod(new_shape) != self.size:
            raise ValueError("Product of current shape is not equal to the product of the new shape!")
        if not self.is_compact():
            raise ValueError("The matrix is not compact!")
        return BackendTensor.make(new_shape, self.compact_strides(new_shape), self._device, self._handle)

    def transpose(self, new_axes=None):
        axes = tuple(new_axes) if new_axes else tuple(reversed(range(self.ndim)))
        return self.permute(axes)

    def permute(self, new_axes):
        axes = list(new_axes)
        return BackendTensor.make(tuple(self._shape[a] for a in axes),
                                  tuple(self._strides[a] for a in axes),
                                  self._device, self._handle, self._offset)

    def broadcast_to(self, new_shape):
        new_shape = tuple(new_shape)
        lead = len(new_shape) - self.ndim
        assert lead >= 0, f"Cannot broadcast {self._shape} to {new_shape}"
        padded = (1,) * lead + tuple(self._shape)
        strides = []
        for axis, (have, want) in enumerate(zip(padded, new_shape)):
            assert have == want or have == 1, f"Dimension mismatch: {have} vs {want}"
            strides.append(0 if (axis < lead or have == 1) else self._strides[axis - lead])
        return BackendTensor.make(new_shape, tuple(strides), self._device, self._handle, self._offset)

    def _norm_slice(self, sl, dim):
        start, stop, step = sl.start, sl.stop, sl.step
        extent = self.shape[dim]
        start = 0 if start is None else (start + extent if start < 0 else start)
        stop = extent if stop is None else (stop + extent if stop < 0 else stop)
        step = 1 if step is None else step
        assert stop > start, "Start must be less than stop"
        assert step > 0, "No support for  negative increments"
        return slice(start, stop, step)

    def __getitem__(self, idxs):
        if not isinstance(idxs, tuple):
            idxs = (idxs,)
        idxs = tuple(self._norm_slice(s, d) if isinstance(s, slice) else slice(s, s + 1, 1)
                     for d, s in enumerate(idxs))
        assert len(idxs) == self.ndim, "Need indexes equal to number of dimensions"
        shape = tuple((s.stop - s.start + s.step - 1) // s.step for s in idxs)
        offset = self._offset + sum(s.start * st for s, st in zip(idxs, self._strides))
        strides = tuple(st * s.step for s, st in zip(idxs, self._strides))
        return BackendTensor.make(shape, strides, self._device, self._handle, offset)

    def __setitem__(self, idxs, other):
        view = self.__getitem__(idxs)
        if isinstance(other, BackendTensor):
            assert prod(view.shape) == prod(other.shape)
            self.device.ewise_setitem(other.compact()._handle, view._handle, view.shape,
                                      view.strides, view._offset)
        else:
            self.device.scalar_setitem(prod(view.shape), other, view._handle, view.shape,
                                       view.strides, view._offset)

    # ---- arithmetic: only `other` is broadcast (:533-542) ------------------------------------
    def ewise_or_scalar(self, other, ewise_func, scalar_func):
        out = BackendTensor.make(self.shape, device=self.device)
        if isinstance(other, BackendTensor):
            if self.shape != other.shape:
                other = other.broadcast_to(self.shape)
            ewise_func(self.compact()._handle, other.compact()._handle, out._handle)
        else:
            scalar_func(self.compact()._handle, other, out._handle)
        return out

    def __add__(self, other):
        return self.ewise_or_scalar(other, self.device.ewise_add, self.device.scalar_add)

    __radd__ = __add__

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, other):
        return other + (-self)

    def __mul__(self, other):
        return self.ewise_or_scalar(other, self.device.ewise_mul, self.device.scalar_mul)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self.ewise_or_scalar(other, self.device.ewise_div, self.device.scalar_div)

    def __neg__(self):
        return self * (-1)

    def __pow__(self, other):
        out = BackendTensor.make(self.shape, device=self.device)
        self.device.scalar_power(self.compact()._handle, other, out._handle)
        return out

    def maximum(self, other):
        return self.ewise_or_scalar(other, self.device.ewise_maximum, self.device.scalar_maximum)

    def __eq__(self, other):
        return self.ewise_or_scalar(other, self.device.ewise_eq, self.device.scalar_eq)

    def __ge__(self, other):
        return self.ewise_or_scalar(other, self.device.ewise_ge, self.device.scalar_ge)

    def __ne__(self, other):
        return 1 - (self == other)

    def __lt__(self, other):
        return 1 - (self >= other)

    def __gt__(self, other):
        return (self >= other) * (self != other)

    def __le__(self, other):
        return 1 - (self > other)

    __hash__ = None

    def _unary(self, fn):
        out = BackendTensor.make(self.shape, device=self.device)
        fn(self.compact()._handle, out._handle)
        return out

    def log(self):
        return self._unary(self.device.ewise_log)

    def exp(self):
        return self._unary(self.device.ewise_exp)

    def tanh(self):
        return self._unary(self.device.ewise_tanh)

    def __matmul__(self, other):
        assert self.ndim == 2 and other.ndim == 2
        assert self.shape[1] == other.shape[0]
        m, n, p = self.shape[0], self.shape[1], other.shape[1]
        out = BackendTensor.make((m, p), device=self.device)
        gemm = getattr(self.device.mod, "gemm", None)
        if gemm is not None:
            # transposed views go straight to the MFMA kernel (no compact() copy, tensor.py:699-716)
            a, ta = _as_gemm_operand(self)
            b, tb = _as_gemm_operand(other)
            gemm(a._handle, b._handle, None, out._handle, m, n, p, ta, tb)
        else:
            self.device.matmul(self.compact()._handle, other.compact()._handle, out._handle, m, n, p)
        return out

    # ---- reductions (single axis; :624-662) ----------------------------------------------------
    def reduce_view_out(self, axis, keepdims=False):
        if isinstance(axis, tuple) and not axis:
            raise ValueError("Empty axis in reduce")
        if axis is None:
            view = self.compact().reshape((1,) * (self.ndim - 1) + (prod(self.shape),))
            out = BackendTensor.make((1,) * (self.ndim if keepdims else 1), device=self.device)
            return view, out
        if isinstance(axis, (tuple, list)):
            assert len(axis) == 1, "Only support reduction over a single axis"
            axis = axis[0]
        view = self.permute(tuple(a for a in range(self.ndim) if a != axis) + (axis,))
        out_shape = (tuple(1 if i == axis else s for i, s in enumerate(self.shape)) if keepdims
                     else tuple(s for i, s in enumerate(self.shape) if i != axis))
        return view, BackendTensor.make(out_shape, device=self.device)

    def _reduce(self, axis, keepdims, prim, op):
        fast = getattr(self.device.mod, "reduce_axis", None)
        if fast is not None and axis is not None and self.is_compact():
            ax = axis[0] if isinstance(axis, (tuple, list)) else axis
            out_shape = (tuple(1 if i == ax else s for i, s in enumerate(self.shape)) if keepdims
                         else tuple(s for i, s in enumerate(self.shape) if i != ax))
            out = BackendTensor.make(out_shape, device=self.device)
            fast(self._handle, out._handle, prod(self.shape[:ax]), self.shape[ax],
                 prod(self.shape[ax + 1:]), op, 1.0)
            return out
        view, out = self.reduce_view_out(axis, keepdims=keepdims)
        prim(view.compact()._handle, out._handle, view.shape[-1])
        return out

    def sum(self, axis=None, keepdims=False):
        return self._reduce(axis, keepdims, self.device.reduce_sum, "sum")

    def max(self, axis=None, keepdims=False):
        return self._reduce(axis, keepdims, self.device.reduce_max, "max")

    def mean(self, axis=None, keepdims=False):
        return self.sum(axis, keepdims=keepdims) / prod(self.shape)   # TOTAL count (:659-662)

    def flip(self, axes):
        assert len(axes) <= len(self.shape)
        strides = list(self.strides)
        offset = self._offset
        for ax in axes:
            offset += (self.shape[ax] - 1) * self.strides[ax]
            strides[ax] = -strides[ax]
        return BackendTensor.make(self.shape, tuple(strides), self._device, self._handle, offset).compact()

    def pad(self, axes):
        assert len(axes) == len(self.shape)
        out = self.device.full(tuple(l + r + n for (l, r), n in zip(axes, self.shape)), 0)
        out[tuple(slice(l, l + n) for (l, _), n in zip(axes, self.shape))] = self
        return out


def _as_gemm_operand(t):
    """(compact tensor, transposed?) for a 2-D view: a plain transpose of a compact matrix is
    passed as-is with the transpose flag; anything else is compacted first."""
    if t.is_compact():
        return t, False
    r, c = t.shape
    if (t._offset == 0 and tuple(t._strides) == (1, r) and t._handle.size == r * c):
        return BackendTensor.make((c, r), None, t._device, t._handle), True
    return t.compact(), False


# ---- module-level helpers (:692-779) ----------------------------------------------------------
def Btensor(a, dtype="float32", device=None):
    return BackendTensor(a, device=device)


def empty(shape, dtype="float32", device=None):
    return (device if device is not None else default_device()).empty(shape, dtype)


def full(shape, fill_value, dtype="float32", device=None):
    return (device if device is not None else default_device()).full(shape, fill_value, dtype)


def zeros(shape, dtype="float32", device=None):
    return full(shape, 0.0, dtype, device)


def ones(shape, dtype="float32", device=None):
    return full(shape, 1.0, dtype, device)


def zeros_like(data):
    return data.device.full(data.shape, 0.0, data.dtype)


def ones_like(data):
    return data.device.full(data.shape, 1.0, data.dtype)


def broadcast_to(array, new_shape):
    return array.broadcast_to(new_shape)


def reshape(array, new_shape):
    return array.reshape(new_shape)


def maximum(a, b):
    return a.maximum(b)


def max(a, axis=None, keepdims=False):
    return a.max(axis, keepdims)


def log(a):
    return a.log()


def exp(a):
    return a.exp()


def tanh(a):
    return a.tanh()


def flip(a, axes):
    return a.flip(axes)


def summation(a, axis=None, keepdims=False):
    return a.sum(axis=axis, keepdims=keepdims)


def mean(a, axis=None, keepdims=False):
    return a.mean(axis=axis, keepdims=keepdims)


def pad(a, axes):
    return a.pad(axes)


def expand_dims(a, axis):
    if abs(axis) > a.ndim:
        raise ValueError(f"axis {axis} is out of bounds for Btensor of dimension {a.ndim}")
    return a.reshape(a.shape[:axis] + (1,) + a.shape[axis:])
