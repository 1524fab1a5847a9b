"""TEST INFRASTRUCTURE — NumPy device module: CPU restatement of the reference's NumPy device
(DeepFlows/backend/backend_tensor.py:64-172 `NumpyBackend`), one method per protocol entry
point, each following the cited reference lines.

One deliberate difference (oracle tier O2): `ewise_setitem` / `scalar_setitem`
SCATTER into the strided view — the semantics of the reference's CUDA device
(backend_src/ndarray_backend_cuda.cu:178-221) — whereas the shipped NumpyBackend writes the
flat prefix of the handle (backend_tensor.py:92-100), which makes pad/im2col/Conv2d/MaxPool2d
compute garbage on its cpu device.  `O1_PREFIX_SETITEM = True` restores the shipped behaviour.
"""
import numpy as np


class NumpyDevice:
    name = "cpu"
    fused = False
    lazy_grad = False
    O1_PREFIX_SETITEM = False
    DT = np.float32          # element type of every array (the reference: float32 only)
    ITEM = 4

    def __init__(self):
        self.calls = {}          # primitive-call histogram
        self.count_calls = False

    def _tick(self, name):
        if self.count_calls:
            self.calls[name] = self.calls.get(name, 0) + 1

    # backend_tensor.py:67-68
    def Array(self, size):
        self._tick("Array")
        return np.zeros(int(size), dtype=self.DT)

    # :70-71
    def fill(self, handle, value):
        self._tick("fill")
        handle[...] = self.DT(value)

    # :73-74
    def from_numpy(self, np_array, handle):
        self._tick("from_numpy")
        handle[:] = np.asarray(np_array).reshape(-1).astype(self.DT)

    def _view(self, handle, shape, strides, offset, writeable=False):
        shape = tuple(int(s) for s in shape)
        strides = tuple(int(s) for s in strides)
        offset = int(offset)
        if len(shape) == 0:
            return handle[offset:offset + 1].reshape(())
        lo = offset + sum((n - 1) * st for n, st in zip(shape, strides) if st < 0)
        hi = offset + sum((n - 1) * st for n, st in zip(shape, strides) if st > 0)
        if any(n == 0 for n in shape):
            return np.zeros(shape, dtype=self.DT)
        if lo < 0 or hi >= handle.size:
            raise IndexError("view exceeds the array")
        if all(st >= 0 for st in strides):
            return np.lib.stride_tricks.as_strided(handle[offset:], shape=shape,
                                                   strides=tuple(self.ITEM * s for s in strides),
                                                   writeable=writeable)
        idx = np.indices(shape, dtype=np.int64)
        lin = offset + sum(idx[d] * strides[d] for d in range(len(shape)))
        return lin  # caller handles the index form

    # :76-90
    def to_numpy(self, handle, shape, strides, offset):
        self._tick("to_numpy")
        v = self._view(handle, shape, strides, offset)
        if v.dtype == np.int64 and v.shape == tuple(int(s) for s in shape) and not np.issubdtype(v.dtype, np.floating):
            return handle[v].copy()
        return np.array(v, dtype=self.DT, copy=True)

    # ndarray_backend_cuda.cu:178-198 (scatter); backend_tensor.py:92-95 when O1_PREFIX_SETITEM
    def ewise_setitem(self, other_handle, view_handle, shape, strides, offset):
        self._tick("ewise_setitem")
        if self.O1_PREFIX_SETITEM:
            view = self.to_numpy(view_handle, shape, strides, offset)
            view[...] = other_handle.reshape(view.shape)
            view_handle[:view.size] = view.reshape(-1)
            return
        v = self._view(view_handle, shape, strides, offset, writeable=True)
        if np.issubdtype(v.dtype, np.floating):
            v[...] = other_handle.reshape(v.shape)
        else:
            view_handle[v.reshape(-1)] = other_handle.reshape(-1)

    # ndarray_backend_cuda.cu:200-221; backend_tensor.py:97-100 when O1_PREFIX_SETITEM
    def scalar_setitem(self, size, scalar, view_handle, shape, strides, offset):
        self._tick("scalar_setitem")
        if self.O1_PREFIX_SETITEM:
            view = self.to_numpy(view_handle, shape, strides, offset)
            view[...] = self.DT(scalar)
            view_handle[:view.size] = view.reshape(-1)
            return
        v = self._view(view_handle, shape, strides, offset, writeable=True)
        if np.issubdtype(v.dtype, np.floating):
            v[...] = self.DT(scalar)
        else:
            view_handle[v.reshape(-1)] = self.DT(scalar)

    # :102-104
    def compact(self, src_handle, dst_handle, shape, strides, offset):
        self._tick("compact")
        dst_handle[:] = self.to_numpy(src_handle, shape, strides, offset).reshape(-1)

    # :106-143
    def ewise_add(self, a, b, out): self._tick("ewise_add"); out[:] = a + b
    def scalar_add(self, a, s, out): self._tick("scalar_add"); out[:] = a + self.DT(s)
    def ewise_mul(self, a, b, out): self._tick("ewise_mul"); out[:] = a * b
    def scalar_mul(self, a, s, out): self._tick("scalar_mul"); out[:] = a * self.DT(s)

    def ewise_div(self, a, b, out):
        self._tick("ewise_div")
        with np.errstate(divide="ignore", invalid="ignore"):
            out[:] = a / b

    def scalar_div(self, a, s, out):
        self._tick("scalar_div")
        with np.errstate(divide="ignore", invalid="ignore"):
            out[:] = a / self.DT(s)

    def scalar_power(self, a, s, out):
        self._tick("scalar_power")
        with np.errstate(divide="ignore", invalid="ignore"):
            out[:] = a ** self.DT(s)

    def ewise_maximum(self, a, b, out): self._tick("ewise_maximum"); out[:] = np.maximum(a, b)
    def scalar_maximum(self, a, s, out): self._tick("scalar_maximum"); out[:] = np.maximum(a, self.DT(s))
    def ewise_eq(self, a, b, out): self._tick("ewise_eq"); out[:] = (a == b).astype(self.DT)
    def scalar_eq(self, a, s, out): self._tick("scalar_eq"); out[:] = (a == self.DT(s)).astype(self.DT)
    def ewise_ge(self, a, b, out): self._tick("ewise_ge"); out[:] = (a >= b).astype(self.DT)
    def scalar_ge(self, a, s, out): self._tick("scalar_ge"); out[:] = (a >= self.DT(s)).astype(self.DT)

    # :145-152
    def ewise_log(self, a, out):
        self._tick("ewise_log")
        with np.errstate(divide="ignore", invalid="ignore"):
            out[:] = np.log(a)

    def ewise_exp(self, a, out):
        self._tick("ewise_exp")
        with np.errstate(over="ignore"):
            out[:] = np.exp(a)

    def ewise_tanh(self, a, out): self._tick("ewise_tanh"); out[:] = np.tanh(a)

    # :154-158
    def matmul(self, a, b, out, m, n, p):
        self._tick("matmul")
        out[:] = (a.reshape(int(m), int(n)) @ b.reshape(int(n), int(p))).reshape(-1)

    # :160-170
    def reduce_sum(self, view_handle, out_handle, last_dim):
        self._tick("reduce_sum")
        last = int(last_dim)
        outer = int(view_handle.size // last)
        out_handle[:outer] = view_handle.reshape(outer, last).sum(axis=1).astype(self.DT)

    def reduce_max(self, view_handle, out_handle, last_dim):
        self._tick("reduce_max")
        last = int(last_dim)
        outer = int(view_handle.size // last)
        out_handle[:outer] = view_handle.reshape(outer, last).max(axis=1).astype(self.DT)


class NumpyDevice64(NumpyDevice):
    """The same device in float64: the reference's algorithm evaluated to double precision, the yardstick the
    tolerance tests use to measure how far an fp32 evaluation (the reference's own, the oracle's, the hip
    device's) sits from the exact result of the same arithmetic (tests/test_error_budget.py)."""
    name = "cpu64"
    DT = np.float64
    ITEM = 8
