"""The `hip` device module: the object a BackendDevice wraps (reference protocol:
DeepFlows/backend/backend_tensor.py:11-28 `BackendDevice.__getattr__` delegates every call
to `mod`; the NumPy module is :66-170, the CUDA pybind module
backend_src/ndarray_backend_cuda.cu:515-717).

`HipModule` exposes
  * the 26 protocol entry points with the reference's names and argument order
    (`Array, fill, from_numpy, to_numpy, compact, ewise_setitem, scalar_setitem,
    ewise_{add,mul,div,maximum,eq,ge}, scalar_{add,mul,div,power,maximum,eq,ge},
    ewise_{log,exp,tanh}, matmul, reduce_{sum,max}`), and
  * the fused hot-path entry points (conv2d_*, bn2d_*, maxpool2d_*, relu_*, softmax_ce_*,
    reduce_to_shape, reduce_axis, broadcast_axis, axpy, adam_step, sgd_step, pack, unpack,
    allreduce_sum, graph capture) that a front-end picks up with `hasattr(device.mod, ...)`.

Every method is a thin ctypes call into libdfhip.so; nothing is computed in Python.
"""
import ctypes
import os
from ctypes import byref, c_char_p, c_double, c_float, c_int, c_int64, c_void_p

import numpy as np

from . import _lib


class HipArray:
    """Flat fp32 device buffer (reference: CudaArray, ndarray_backend_cuda.cu:48-83).
    `.size` (elements) is part of the protocol (backend_tensor.py:304-308)."""

    __slots__ = ("ptr", "size", "_free", "__weakref__")

    def __init__(self, size, lib):
        size = int(size)
        if size < 0:
            raise ValueError("Array size must be >= 0")
        out = c_void_p()
        _lib.check(lib.dfhip_malloc(size, byref(out)))
        self.ptr = out.value
        self.size = size
        self._free = lib.dfhip_free

    def __del__(self):
        p = getattr(self, "ptr", None)
        if p:
            try:
                self._free(p)
            except Exception:
                pass
            self.ptr = None

    def __repr__(self):
        return f"HipArray(size={self.size}, ptr=0x{(self.ptr or 0):x})"


class DeferredHipArray(HipArray):
    """A device array whose contents one more step has to complete: `finish` (a callable that enqueues that step) runs the
    first time anything asks for the array's address -- every use of an array goes through `.ptr` -- unless whoever owes the
    step has delivered it meanwhile (`settled()`).  Used for the loss of the fused classifier head: its launch leaves row
    losses, their mean is added up by the backward pass's slab-sum launch, or by one small launch when the loss is read
    first (dfhip_head_ce_forward in include/dfhip.h says why the launch itself does not finish it).  `.raw_ptr` is the
    address without that side effect (for the launches that WRITE the array)."""

    __slots__ = ("_raw", "_finish")

    def __init__(self, size, lib):
        self._finish = None
        HipArray.__init__(self, size, lib)

    @property
    def ptr(self):
        f = self._finish
        if f is not None:
            self._finish = None
            f()
        return self._raw

    @ptr.setter
    def ptr(self, value):
        self._raw = value

    @property
    def raw_ptr(self):
        return self._raw

    def owe(self, finish):
        self._finish = finish

    def settled(self):
        self._finish = None

    def __del__(self):
        self._finish = None                         # (nothing is launched for an array nobody will read)
        HipArray.__del__(self)


def _dims(seq):
    n = len(seq)
    return n, (c_int64 * max(n, 1))(*[int(v) for v in seq])


def _ptr_table(handles):
    return (c_void_p * len(handles))(*[h.ptr if h is not None else None for h in handles])


class HipGraph:
    """An instantiated hipGraph of everything enqueued between capture_begin/capture_end."""

    def __init__(self, mod, handle, keep):
        self._mod = mod
        self._h = handle
        self._keep = keep  # python objects whose device buffers the graph reads/writes

    def replay(self):
        _lib.check(self._mod._lib.dfhip_graph_launch(self._h))

    def destroy(self):
        if self._h:
            self._mod._lib.dfhip_graph_destroy(self._h)
            self._h = None
            self._keep = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HipModule:
    name = "hip"
    fused = True        # front-end: use the fused entry points
    lazy_grad = True    # front-end: do not allocate a zero .grad per autograd node

    def __init__(self, device_id=-1):
        self._lib = _lib.load()
        _lib.check(self._lib.dfhip_init(int(device_id)))
        buf = ctypes.create_string_buffer(256)
        _lib.check(self._lib.dfhip_device_name(buf, 256))
        self.device_name = buf.value.decode()
        self.conv_dgrad_mode = "overwrite"   # reference-exact col2im; "accumulate" = true gradient

    # Opt-in reduced precision of the matrix products of the layer1-2 forward convolutions and of the grouped convolution
    # weight gradients (outside the 1e-4 parity contract, never the default): "f32" = the reference's
    # arithmetic, "bf16" = bf16 operands, fp32 accumulation (dfhip_set_conv_matmul_dtype).  Everything else computes in fp32
    # either way.
    @property
    def conv_matmul_dtype(self):
        return "bf16" if self._lib.dfhip_conv_matmul_dtype() == 1 else "f32"

    @conv_matmul_dtype.setter
    def conv_matmul_dtype(self, value):
        if value not in ("f32", "bf16"):
            raise ValueError("conv_matmul_dtype is 'f32' or 'bf16'")
        _lib.check(self._lib.dfhip_set_conv_matmul_dtype(1 if value == "bf16" else 0))

    # ---- memory ---------------------------------------------------------------------------
    def Array(self, size):
        return HipArray(size, self._lib)

    def fill(self, handle, value):
        _lib.check(self._lib.dfhip_fill(handle.ptr, handle.size, float(value)))

    def from_numpy(self, np_array, handle):
        # any dtype / shape, contiguous (nn/init.py:16 passes float64); pybind cast it to float32
        a = np.ascontiguousarray(np_array, dtype=np.float32).reshape(-1)
        if a.size != handle.size:
            raise ValueError(f"from_numpy: array has {a.size} elements, handle has {handle.size}")
        _lib.check(self._lib.dfhip_h2d(a.ctypes.data, handle.ptr, a.size))

    def upload_bernoulli_mask(self, handle, keep):
        """The Dropout mask np.random.binomial(1, keep, handle.size) would draw next, as float32, generated in C
        straight into pinned memory and sent to `handle` (NumPy's global generator is advanced exactly as that call
        would advance it).  False when the generator is not the legacy MT19937: the caller draws with NumPy."""
        st = np.random.get_state()
        if st[0] != "MT19937":
            return False
        key = np.array(st[1], dtype=np.uint32, copy=True)
        pos = ctypes.c_int(int(st[2]))
        _lib.check(self._lib.dfhip_upload_bernoulli_mask(key.ctypes.data, ctypes.byref(pos), float(keep), handle.size,
                                                         handle.ptr))
        np.random.set_state(("MT19937", key, pos.value, st[3], st[4]))
        return True

    def upload_bernoulli_masks(self, feeds):
        """Every Dropout mask of a step -- feeds = [(array, keep), ...] in forward order -- in one call: drawn one after
        the other from NumPy's global generator as upload_bernoulli_mask draws one, sent beside the step that is running
        and turned into floats by one launch.  False when the generator is not the legacy MT19937."""
        st = np.random.get_state()
        if st[0] != "MT19937":
            return False
        cnt = len(feeds)
        key = np.array(st[1], dtype=np.uint32, copy=True)
        pos = ctypes.c_int(int(st[2]))
        keeps = (ctypes.c_double * cnt)(*[float(k) for _, k in feeds])
        ns = (ctypes.c_int64 * cnt)(*[int(m._handle.size) for m, _ in feeds])
        dsts = (ctypes.c_void_p * cnt)(*[int(m._handle.ptr) for m, _ in feeds])
        _lib.check(self._lib.dfhip_upload_bernoulli_masks(key.ctypes.data, ctypes.byref(pos), cnt, keeps, ns, dsts))
        np.random.set_state(("MT19937", key, pos.value, st[3], st[4]))
        return True

    def to_numpy(self, handle, shape, strides, offset):
        shape = tuple(int(s) for s in shape)
        strides = tuple(int(s) for s in strides)
        offset = int(offset)
        if any(s == 0 for s in shape):
            return np.zeros(shape, dtype=np.float32)
        lo = offset + sum((n - 1) * st for n, st in zip(shape, strides) if st < 0)
        hi = offset + sum((n - 1) * st for n, st in zip(shape, strides) if st > 0)
        if lo < 0 or hi >= handle.size:
            raise IndexError("to_numpy: view exceeds the array")
        span = np.empty(hi - lo + 1, dtype=np.float32)
        _lib.check(self._lib.dfhip_d2h(handle.ptr + 4 * lo, span.ctypes.data, span.size))
        view = np.lib.stride_tricks.as_strided(span[offset - lo:], shape=shape,
                                               strides=tuple(4 * s for s in strides))
        return np.array(view, dtype=np.float32, copy=True)

    # (parameter names as the reference's device modules spell them -- backend_tensor.py:66-170, pinned by
    # tests/golden/protocol_signatures.json -- so that a keyword call is a drop-in too)
    def compact(self, src_handle, dst_handle, shape, strides, offset):
        nd, sh = _dims(shape)
        _, st = _dims(strides)
        _lib.check(self._lib.dfhip_compact(src_handle.ptr, dst_handle.ptr, dst_handle.size, nd, sh, st, int(offset)))

    def ewise_setitem(self, other_handle, view_handle, shape, strides, offset):
        nd, sh = _dims(shape)
        _, st = _dims(strides)
        _lib.check(self._lib.dfhip_ewise_setitem(other_handle.ptr, view_handle.ptr, other_handle.size, nd, sh, st,
                                                 int(offset)))

    def scalar_setitem(self, size, scalar, view_handle, shape, strides, offset):
        nd, sh = _dims(shape)
        _, st = _dims(strides)
        _lib.check(self._lib.dfhip_scalar_setitem(int(size), float(scalar), view_handle.ptr, nd, sh, st, int(offset)))

    # ---- elementwise ----------------------------------------------------------------------
    def _binary(self, fn, a, b, out):
        if a.size != out.size or b.size != out.size:
            raise ValueError("elementwise operands must have the same size")
        _lib.check(fn(a.ptr, b.ptr, out.ptr, out.size))

    def _scalar(self, fn, a, s, out):
        if a.size != out.size:
            raise ValueError("elementwise operands must have the same size")
        _lib.check(fn(a.ptr, float(s), out.ptr, out.size))

    def _unary(self, fn, a, out):
        if a.size != out.size:
            raise ValueError("elementwise operands must have the same size")
        _lib.check(fn(a.ptr, out.ptr, out.size))

    def ewise_add(self, a, b, out): self._binary(self._lib.dfhip_ewise_add, a, b, out)
    def ewise_mul(self, a, b, out): self._binary(self._lib.dfhip_ewise_mul, a, b, out)
    def ewise_div(self, a, b, out): self._binary(self._lib.dfhip_ewise_div, a, b, out)
    def ewise_maximum(self, a, b, out): self._binary(self._lib.dfhip_ewise_maximum, a, b, out)
    def ewise_eq(self, a, b, out): self._binary(self._lib.dfhip_ewise_eq, a, b, out)
    def ewise_ge(self, a, b, out): self._binary(self._lib.dfhip_ewise_ge, a, b, out)

    def scalar_add(self, a, s, out): self._scalar(self._lib.dfhip_scalar_add, a, s, out)
    def scalar_mul(self, a, s, out): self._scalar(self._lib.dfhip_scalar_mul, a, s, out)
    def scalar_div(self, a, s, out): self._scalar(self._lib.dfhip_scalar_div, a, s, out)
    def scalar_power(self, a, s, out): self._scalar(self._lib.dfhip_scalar_power, a, s, out)
    def scalar_maximum(self, a, s, out): self._scalar(self._lib.dfhip_scalar_maximum, a, s, out)
    def scalar_eq(self, a, s, out): self._scalar(self._lib.dfhip_scalar_eq, a, s, out)
    def scalar_ge(self, a, s, out): self._scalar(self._lib.dfhip_scalar_ge, a, s, out)

    def ewise_log(self, a, out): self._unary(self._lib.dfhip_ewise_log, a, out)
    def ewise_exp(self, a, out): self._unary(self._lib.dfhip_ewise_exp, a, out)
    def ewise_tanh(self, a, out): self._unary(self._lib.dfhip_ewise_tanh, a, out)

    # ---- contraction / reduction ------------------------------------------------------------
    def matmul(self, a, b, out, m, n, p):
        m, n, p = int(m), int(n), int(p)
        if a.size != m * n or b.size != n * p or out.size != m * p:
            raise ValueError("matmul: array sizes do not match (m, n, p)")
        _lib.check(self._lib.dfhip_matmul(a.ptr, b.ptr, out.ptr, m, n, p))

    def reduce_sum(self, view_handle, out_handle, last_dim):
        _lib.check(self._lib.dfhip_reduce_sum(view_handle.ptr, out_handle.ptr, view_handle.size // int(last_dim), int(last_dim)))

    def reduce_max(self, view_handle, out_handle, last_dim):
        _lib.check(self._lib.dfhip_reduce_max(view_handle.ptr, out_handle.ptr, view_handle.size // int(last_dim), int(last_dim)))

    # ---- fused hot path ---------------------------------------------------------------------
    def gemm(self, a, b, bias, out, m, n, p, ta=False, tb=False):
        _lib.check(self._lib.dfhip_gemm(a.ptr, b.ptr, bias.ptr if bias is not None else None,
                                        out.ptr, int(m), int(n), int(p), int(ta), int(tb)))

    def conv2d_forward(self, x, w, bias, y, N, C, H, W, O, k, stride, pad):
        _lib.check(self._lib.dfhip_conv2d_forward(x.ptr, w.ptr, bias.ptr if bias is not None else None,
                                                  y.ptr, N, C, H, W, O, k, stride, pad))

    def conv2d_forward_stats(self, x, w, bias, y, N, C, H, W, O, k, stride, pad, partials, cap_rows):
        """conv2d_forward that also fills `partials` (cap_rows x O x 3) with BatchNorm partials of y;
        returns the number of rows written (0: the shape took a kernel without that epilogue)."""
        rows = ctypes.c_int(0)
        _lib.check(self._lib.dfhip_conv2d_forward_stats(
            x.ptr, w.ptr, bias.ptr if bias is not None else None, y.ptr, N, C, H, W, O, k, stride, pad,
            partials.ptr, int(cap_rows), ctypes.byref(rows)))
        return int(rows.value)

    def bn2d_forward_partials(self, x, partials, rows, gamma, beta, residual, y, save_mean, save_var,
                              running_mean, running_var, N, C, HW, eps, momentum):
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_bn2d_forward_partials(
            x.ptr, partials.ptr, int(rows), g(gamma), g(beta), g(residual), y.ptr, g(save_mean), g(save_var),
            g(running_mean), g(running_var), N, C, HW, float(eps), float(momentum)))

    def conv2d_wgrad(self, x, dy, dw, N, C, H, W, O, k, stride, pad):
        _lib.check(self._lib.dfhip_conv2d_wgrad(x.ptr, dy.ptr, dw.ptr, N, C, H, W, O, k, stride, pad))
        if self._wgrad_hold is not None:
            # the call may only have been queued: x and dy must outlive the grouped launch
            self._wgrad_hold.append((x, dy, dw))

    def conv2d_wgrad_affine(self, x, affine, dy, dw, N, C, H, W, O, k, stride, pad):
        _lib.check(self._lib.dfhip_conv2d_wgrad_affine(x.ptr, affine.ptr, dy.ptr, dw.ptr, N, C, H, W, O, k, stride,
                                                       pad))
        if self._wgrad_hold is not None:
            self._wgrad_hold.append((x, affine, dy, dw))

    def conv2d_forward_bn_input(self, x, partials_in, rows_in, gamma, beta, eps, momentum, save_mean, save_var,
                                running_mean, running_var, affine, w, bias, y, N, C, H, W, O, k, stride, pad,
                                partials, cap_rows):
        """conv2d(batch_norm_train(x)) with the normalisation applied on the operand loads; returns (done, rows):
        done False = the shape has no such kernel (nothing was launched), rows = BatchNorm partial rows of y."""
        g = lambda h: h.ptr if h is not None else None
        rows, done = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.dfhip_conv2d_forward_bn_input(
            x.ptr, partials_in.ptr, int(rows_in), g(gamma), g(beta), float(eps), float(momentum), save_mean.ptr,
            save_var.ptr, g(running_mean), g(running_var), affine.ptr, w.ptr, g(bias), y.ptr, N, C, H, W, O, k,
            stride, pad, g(partials), int(cap_rows), ctypes.byref(rows), ctypes.byref(done)))
        return bool(done.value), int(rows.value)

    # Grouped weight gradients (dfhip_wgrad_group_begin / _end): Tensor.backward() brackets its
    # sweep with these, so the weight gradients of all layers run as one launch when it ends.
    _wgrad_hold = None

    def wgrad_group_begin(self):
        if not self.wgrad_grouping:
            return False
        _lib.check(self._lib.dfhip_wgrad_group_begin())
        self._wgrad_hold = []
        return True

    def wgrad_group_end(self):
        try:
            _lib.check(self._lib.dfhip_wgrad_group_end())
        finally:
            self._wgrad_hold = None

    wgrad_grouping = os.environ.get("DF_WGRAD_GROUP", "1") != "0"

    def conv2d_forward_s2_pair(self, x, w, wd, y, yd, N, C, H, W, O, partials, partials_d, cap_rows):
        """y = conv3x3/s2/p1(x, w), yd = conv1x1/s2/p0(x, wd) as one launch; returns (done, rows of BatchNorm partials)."""
        g = lambda h: h.ptr if h is not None else None
        rows, done = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.dfhip_conv2d_forward_s2_pair(x.ptr, w.ptr, wd.ptr, y.ptr, yd.ptr, N, C, H, W, O, g(partials),
                                                          g(partials_d), int(cap_rows), ctypes.byref(rows),
                                                          ctypes.byref(done)))
        return bool(done.value), int(rows.value)

    def conv2d_dgrad_pair(self, dy, w, dy2, w2, add, dx, N, C, H, W, O, mode=None):
        """dx = dgrad(3x3 s2 p1; dy, w) + dgrad(1x1 s2 p0; dy2, w2) (+ add) as one launch; False = no such kernel."""
        mode = self.conv_dgrad_mode if mode is None else mode
        m = {"overwrite": 0, "accumulate": 1}[mode]
        done = ctypes.c_int(0)
        _lib.check(self._lib.dfhip_conv2d_dgrad_pair(dy.ptr, w.ptr, dy2.ptr, w2.ptr, add.ptr if add is not None else None,
                                                     dx.ptr, N, C, H, W, O, m, ctypes.byref(done)))
        return bool(done.value)

    def conv2d_dgrad(self, dy, w, dx, N, C, H, W, O, k, stride, pad, mode=None):
        mode = self.conv_dgrad_mode if mode is None else mode
        m = {"overwrite": 0, "accumulate": 1}[mode]
        _lib.check(self._lib.dfhip_conv2d_dgrad(dy.ptr, w.ptr, dx.ptr, N, C, H, W, O, k, stride, pad, m))

    def conv2d_dgrad_add(self, dy, w, add, dx, N, C, H, W, O, k, stride, pad, mode=None):
        """dx = conv2d_dgrad(dy, w) + add (gradient accumulation in the kernel's epilogue)."""
        mode = self.conv_dgrad_mode if mode is None else mode
        m = {"overwrite": 0, "accumulate": 1}[mode]
        _lib.check(self._lib.dfhip_conv2d_dgrad_add(dy.ptr, w.ptr, add.ptr, dx.ptr, N, C, H, W, O, k, stride, pad, m))

    # ---- BatchNorm2d backward without a launch of its own (include/dfhip.h: dfhip_bn_bwd_in / dfhip_bn_stats_out) ----
    _bn_supported = None

    def conv2d_bn_bwd_supported(self, N, C, H, W, O, k, stride, pad, pair=False):
        """(dgrad_ok, wgrad_ok, stats_rows): the convolution's input / weight gradient have kernels that apply a pending
        BatchNorm backward on their operand loads (overwrite mode); the rows that input gradient leaves behind for the
        BatchNorm layers its output feeds.  Cached per shape."""
        if self._bn_supported is None:
            self._bn_supported = {}
        key = (N, C, H, W, O, k, stride, pad, bool(pair))
        hit = self._bn_supported.get(key)
        if hit is None:
            a, b, r = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
            _lib.check(self._lib.dfhip_conv2d_bn_bwd_supported(N, C, H, W, O, k, stride, pad, 1 if pair else 0,
                                                               ctypes.byref(a), ctypes.byref(b), ctypes.byref(r)))
            hit = self._bn_supported[key] = (bool(a.value), bool(b.value), int(r.value))
        return hit

    @staticmethod
    def _bn_in(p):
        """ctypes descriptor of a pending BatchNorm backward `p` (attributes z, rows, nrows, K, kidx, mean, var, gamma,
        eps, m, dgamma, dbeta, table: HipArray handles / numbers); None -> NULL."""
        if p is None:
            return None
        g = lambda h: h.ptr if h is not None else None
        return _lib.BnBwdIn(p.z.ptr, p.rows.ptr, int(p.nrows), int(p.K), int(p.kidx), p.mean.ptr, p.var.ptr, g(p.gamma),
                            float(p.eps), 1.0 / float(p.m), g(p.dgamma), g(p.dbeta), g(p.table))

    @staticmethod
    def _bn_out(targets, rows, cap_rows):
        if not targets:
            return None
        so = _lib.BnStatsOut()
        so.n = len(targets)
        for j, (z, mean) in enumerate(targets):
            so.z[j] = z.ptr
            so.mean[j] = mean.ptr
        so.rows = rows.ptr
        so.cap_rows = int(cap_rows)
        so.nrows = 0
        return so

    def conv2d_dgrad_bn(self, dy, w, add, dx, N, C, H, W, O, k, stride, pad, pending=None, targets=None, rows=None,
                        cap_rows=0):
        """overwrite-mode dx = dgrad(dy', w) (+ add) where dy' is the pending BatchNorm backward of dy (`pending`) or dy
        itself, leaving the partial rows of the BatchNorm layers `targets` = [(z, mean), ...] in `rows`; returns the
        number of rows written."""
        bi, so = self._bn_in(pending), self._bn_out(targets, rows, cap_rows)
        _lib.check(self._lib.dfhip_conv2d_dgrad_bn(dy.ptr, w.ptr, add.ptr if add is not None else None, dx.ptr, N, C, H, W,
                                                   O, k, stride, pad, ctypes.byref(bi) if bi is not None else None,
                                                   ctypes.byref(so) if so is not None else None))
        return int(so.nrows) if so is not None else 0

    def conv2d_dgrad_pair_bn(self, dy, w, dy2, w2, add, dx, N, C, H, W, O, pending=None, pending2=None, targets=None,
                             rows=None, cap_rows=0):
        bi, bi2, so = self._bn_in(pending), self._bn_in(pending2), self._bn_out(targets, rows, cap_rows)
        ref = lambda s: ctypes.byref(s) if s is not None else None
        _lib.check(self._lib.dfhip_conv2d_dgrad_pair_bn(dy.ptr, w.ptr, dy2.ptr, w2.ptr, add.ptr if add is not None else None,
                                                        dx.ptr, N, C, H, W, O, ref(bi), ref(bi2), ref(so)))
        return int(so.nrows) if so is not None else 0

    def conv2d_wgrad_bn(self, x, affine, dy, z, table, dw, N, C, H, W, O, k, stride, pad):
        """the (grouped) weight gradient with dy' = pending BatchNorm backward of dy formed on the operand loads; False =
        nothing was queued (not inside a group, or no such kernel for the shape)."""
        done = ctypes.c_int(0)
        _lib.check(self._lib.dfhip_conv2d_wgrad_bn(x.ptr, affine.ptr if affine is not None else None, dy.ptr, z.ptr,
                                                   table.ptr, dw.ptr, N, C, H, W, O, k, stride, pad, ctypes.byref(done)))
        if done.value and self._wgrad_hold is not None:
            self._wgrad_hold.append((x, affine, dy, z, table, dw))
        return bool(done.value)

    def bn_bwd_table(self, pending, O):
        bi = self._bn_in(pending)
        _lib.check(self._lib.dfhip_bn_bwd_table(ctypes.byref(bi), int(O)))

    def bn2d_backward_apply_table(self, dy, z, table, dz, N, O, HW):
        _lib.check(self._lib.dfhip_bn2d_backward_apply_table(dy.ptr, z.ptr, table.ptr, dz.ptr, N, O, HW))

    def bn2d_forward_train(self, x, gamma, beta, y, save_mean, save_var, running_mean, running_var,
                           N, C, HW, eps, momentum):
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_bn2d_forward_train(x.ptr, g(gamma), g(beta), y.ptr, g(save_mean),
                                                      g(save_var), g(running_mean), g(running_var),
                                                      N, C, HW, float(eps), float(momentum)))

    def bn2d_forward_eval(self, x, gamma, beta, y, running_mean, running_var, N, C, HW, eps):
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_bn2d_forward_eval(x.ptr, g(gamma), g(beta), y.ptr, running_mean.ptr,
                                                     running_var.ptr, N, C, HW, float(eps)))

    def bn2d_backward(self, dy, x, gamma, save_mean, save_var, dx, dgamma, dbeta, N, C, HW, eps):
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_bn2d_backward(dy.ptr, x.ptr, g(gamma), save_mean.ptr, save_var.ptr,
                                                 g(dx), g(dgamma), g(dbeta), N, C, HW, float(eps)))

    def linear_backward(self, x, w, dy, dx, dw, db, m, k, p):
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_linear_backward(g(x), g(w), dy.ptr, g(dx), g(dw), g(db), int(m), int(k), int(p)))

    def linear_relu_forward(self, x, w, bias, z, r, m, k, p):
        """relu(x.W + b) in one launch (z = the pre-activation, r = the output); False = too large, nothing launched."""
        done = ctypes.c_int(0)
        _lib.check(self._lib.dfhip_linear_relu_forward(x.ptr, w.ptr, bias.ptr if bias is not None else None, z.ptr, r.ptr,
                                                       int(m), int(k), int(p), ctypes.byref(done)))
        return bool(done.value)

    def linear_relu_backward(self, x, w, z, r, dr, dx, dw, db, m, k, p):
        g = lambda h: h.ptr if h is not None else None
        done = ctypes.c_int(0)
        _lib.check(self._lib.dfhip_linear_relu_backward(g(x), g(w), z.ptr, r.ptr, dr.ptr, g(dx), g(dw), g(db), int(m),
                                                        int(k), int(p), ctypes.byref(done)))
        return bool(done.value)

    def bn2d_forward_partials_pair(self, a, b, y, N, C, HW):
        """a, b = (x, partials, rows, gamma, beta, save_mean, save_var, running_mean, running_var, eps, momentum)."""
        g = lambda h: h.ptr if h is not None else None
        flat = []
        for (x, part, rows, gamma, beta, sm, sv, rm, rv, eps, mom) in (a, b):
            flat += [x.ptr, part.ptr, int(rows), g(gamma), g(beta), g(sm), g(sv), g(rm), g(rv), float(eps), float(mom)]
        _lib.check(self._lib.dfhip_bn2d_forward_partials_pair(*flat, y.ptr, N, C, HW))

    def bn2d_backward_pair(self, dy, a, b, N, C, HW):
        """a, b = (x, gamma, save_mean, save_var, dx, dgamma, dbeta, eps)."""
        g = lambda h: h.ptr if h is not None else None
        flat = []
        for (x, gamma, sm, sv, dx, dg, db, eps) in (a, b):
            flat += [x.ptr, g(gamma), sm.ptr, sv.ptr, g(dx), g(dg), g(db), float(eps)]
        _lib.check(self._lib.dfhip_bn2d_backward_pair(dy.ptr, *flat, N, C, HW))

    def stem_backward(self, x, z, y, g, gamma, beta, save_mean, save_var, dw, dgamma, dbeta, N, C, H, W, O, k, stride,
                      pad, eps):
        """dgamma, dbeta and the convolution's dw of pool(relu(bn(conv(x)))) without the full-resolution gradient
        tensor; False = no such kernel for the shape (nothing was launched)."""
        gp = lambda h: h.ptr if h is not None else None
        done = ctypes.c_int(0)
        _lib.check(self._lib.dfhip_stem_backward(x.ptr, z.ptr, y.ptr, g.ptr, gp(gamma), gp(beta), save_mean.ptr,
                                                 save_var.ptr, dw.ptr, gp(dgamma), gp(dbeta), N, C, H, W, O, k, stride,
                                                 pad, float(eps), ctypes.byref(done)))
        return bool(done.value)

    def stem_forward_noz(self, x, w, bias, gamma, beta, y, save_mean, save_var, running_mean, running_var, N, C, H, W, O,
                         k, stride, pad, eps, momentum):
        """pool(relu(bn(conv(x)))) of the CIFAR stems without the full-resolution convolution output; False = the shape
        has no such kernel (nothing was launched)."""
        gp = lambda h: h.ptr if h is not None else None
        done = ctypes.c_int(0)
        _lib.check(self._lib.dfhip_stem_forward_noz(x.ptr, w.ptr, gp(bias), gp(gamma), gp(beta), y.ptr, save_mean.ptr,
                                                    save_var.ptr, gp(running_mean), gp(running_var), N, C, H, W, O, k,
                                                    stride, pad, float(eps), float(momentum), ctypes.byref(done)))
        return bool(done.value)

    def stem_backward_noz(self, x, w, bias, y, g, gamma, beta, save_mean, save_var, dw, dbias, dgamma, dbeta, N, C, H, W,
                          O, k, stride, pad, eps):
        gp = lambda h: h.ptr if h is not None else None
        done = ctypes.c_int(0)
        _lib.check(self._lib.dfhip_stem_backward_noz(x.ptr, w.ptr, gp(bias), y.ptr, g.ptr, gp(gamma), gp(beta),
                                                     save_mean.ptr, save_var.ptr, dw.ptr, gp(dbias), gp(dgamma),
                                                     gp(dbeta), N, C, H, W, O, k, stride, pad, float(eps),
                                                     ctypes.byref(done)))
        return bool(done.value)

    def bn_relu_maxpool2_forward_partials(self, x, partials, rows, gamma, beta, y, save_mean, save_var, running_mean,
                                          running_var, N, C, H, W, eps, momentum):
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_bn_relu_maxpool2_forward_partials(
            x.ptr, partials.ptr, int(rows), g(gamma), g(beta), y.ptr, save_mean.ptr, save_var.ptr, g(running_mean),
            g(running_var), N, C, H, W, float(eps), float(momentum)))

    def bn_relu_maxpool2_forward(self, x, gamma, beta, y, save_mean, save_var, running_mean, running_var,
                                 N, C, H, W, eps, momentum):
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_bn_relu_maxpool2_forward(x.ptr, g(gamma), g(beta), y.ptr, save_mean.ptr,
                                                            save_var.ptr, g(running_mean), g(running_var),
                                                            N, C, H, W, float(eps), float(momentum)))

    def bn_relu_maxpool2_backward(self, x, y, grad, gamma, beta, save_mean, save_var, dx, dgamma, dbeta,
                                  N, C, H, W, eps):
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_bn_relu_maxpool2_backward(x.ptr, y.ptr, grad.ptr, g(gamma), g(beta),
                                                             save_mean.ptr, save_var.ptr, g(dx), g(dgamma),
                                                             g(dbeta), N, C, H, W, float(eps)))

    # ---- input pipeline ----------------------------------------------------------------------------
    def upload_async(self, host_array, dst, slot, dst_offset_floats=0):
        """Enqueue host_array (any dtype, C-contiguous) into dst[dst_offset_floats:] through pinned slot `slot`
        on the copy stream; returns at once."""
        a = np.ascontiguousarray(host_array)
        _lib.check(self._lib.dfhip_upload_async(a.ctypes.data, dst.ptr + 4 * int(dst_offset_floats), a.nbytes,
                                                int(slot)))

    def upload_wait(self, slot):
        _lib.check(self._lib.dfhip_upload_wait(int(slot)))

    def upload_release(self, slot):
        """the kernels that read slot's last upload are enqueued: the next upload through it may start behind them"""
        _lib.check(self._lib.dfhip_upload_release(int(slot)))

    def augment_batch(self, src, out, decisions, N, C, H, W, pad, src_offset_floats=0, dec_offset_floats=0):
        _lib.check(self._lib.dfhip_augment_batch(src.ptr + 4 * int(src_offset_floats), out.ptr,
                                                 decisions.ptr + 4 * int(dec_offset_floats), N, C, H, W, pad))

    def onehot_smooth(self, labels, out, N, K, eps, labels_offset_floats=0):
        _lib.check(self._lib.dfhip_onehot_smooth(labels.ptr + 4 * int(labels_offset_floats), out.ptr, N, K,
                                                 float(eps)))

    def maxpool2d_forward(self, x, y, NC, H, W, k):
        _lib.check(self._lib.dfhip_maxpool2d_forward(x.ptr, y.ptr, NC, H, W, k))

    def maxpool2d_backward(self, x, y, dy, dx, NC, H, W, k):
        _lib.check(self._lib.dfhip_maxpool2d_backward(x.ptr, y.ptr, dy.ptr, dx.ptr, NC, H, W, k))

    def relu_forward(self, x, y):
        _lib.check(self._lib.dfhip_relu_forward(x.ptr, y.ptr, y.size))

    def relu_backward(self, x, y, g, dx):
        _lib.check(self._lib.dfhip_relu_backward(x.ptr, y.ptr, g.ptr, dx.ptr, dx.size))

    def softmax_ce_forward(self, logits, target, loss, rows, cols):
        _lib.check(self._lib.dfhip_softmax_ce_forward(logits.ptr, target.ptr, loss.ptr, rows, cols))

    def softmax_ce_forward_grad(self, logits, target, loss, dlogits, rows, cols):
        """loss + d loss / d logits for an upstream gradient of 1; False when only the loss was computed."""
        done = ctypes.c_int(0)
        _lib.check(self._lib.dfhip_softmax_ce_forward_grad(logits.ptr, target.ptr, loss.ptr, dlogits.ptr, rows, cols,
                                                           ctypes.byref(done)))
        return bool(done.value)

    def pool_linear_forward(self, x, w, b, pooled, y, N, C, HW, P, scale):
        _lib.check(self._lib.dfhip_pool_linear_forward(x.ptr, w.ptr, b.ptr if b is not None else None, pooled.ptr,
                                                       y.ptr, N, C, HW, P, float(scale)))

    def pool_linear_backward(self, dy, pooled, w, dx, dw, db, N, C, HW, P, gscale):
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_pool_linear_backward(dy.ptr, pooled.ptr, w.ptr, g(dx), g(dw), g(db), N, C, HW, P,
                                                        float(gscale)))

    def head_ce_forward(self, x, w, b, target, pooled, logits, loss, dlogits, dx, rowloss, N, C, HW, P, scale, gscale):
        """classifier head + cross-entropy + everything of their backward pass that belongs to one image, one launch; False
        when the shape is outside the kernel (nothing was launched)"""
        done = ctypes.c_int(0)
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_head_ce_forward(x.ptr, w.ptr, g(b), target.ptr, pooled.ptr, logits.ptr,
                                                   getattr(loss, "raw_ptr", None) or loss.ptr,
                                                   dlogits.ptr, g(dx), rowloss.ptr, N, C, HW, P, float(scale), float(gscale),
                                                   ctypes.byref(done)))
        return bool(done.value)

    def droplin_ce_forward(self, x, mask, keep, w, b, target, xd, logits, loss, dlogits, dx, rowloss, N, C, P):
        """Dropout + classifier + cross-entropy + everything of their backward pass that belongs to one row, one launch;
        False when the shape is outside the kernel (nothing was launched)"""
        done = ctypes.c_int(0)
        g = lambda h: h.ptr if h is not None else None
        _lib.check(self._lib.dfhip_droplin_ce_forward(x.ptr, mask.ptr, float(keep), w.ptr, g(b), target.ptr, xd.ptr, logits.ptr,
                                                      getattr(loss, "raw_ptr", None) or loss.ptr, dlogits.ptr, g(dx),
                                                      rowloss.ptr, N, C, P, ctypes.byref(done)))
        return bool(done.value)

    def head_grads(self, dlogits, pooled, dw, db, N, C, P, rowloss=None, loss=None):
        """the sums over the images of the head's backward pass; with (rowloss, loss) also the mean of the row losses"""
        g = lambda h: h.ptr if h is not None else None
        raw = lambda h: None if h is None else getattr(h, "raw_ptr", None) or h.ptr
        _lib.check(self._lib.dfhip_head_grads(dlogits.ptr, pooled.ptr, g(dw), g(db), g(rowloss), raw(loss), N, C, P))
        if loss is not None and hasattr(loss, "settled"):
            loss.settled()

    def mlp2_ce_forward(self, x, w1, b1, w2, b2, target, zpre, h, logits, dlogits, dz1, rowloss, M, K, HD, P):
        """fc1 -> relu -> fc2 -> cross-entropy and the per-row half of their backward pass, one launch; False when the shape
        is outside the kernel (nothing was launched)"""
        done = ctypes.c_int(0)
        g = lambda a: a.ptr if a is not None else None
        _lib.check(self._lib.dfhip_mlp2_ce_forward(x.ptr, w1.ptr, g(b1), w2.ptr, g(b2), target.ptr, g(zpre), h.ptr, logits.ptr,
                                                   dlogits.ptr, dz1.ptr, rowloss.ptr, M, K, HD, P, ctypes.byref(done)))
        return bool(done.value)

    def mlp2_ce_backward(self, x, h, dz1, dlogits, dw1, db1, dw2, db2, rowloss, loss, M, K, HD, P):
        g = lambda a: a.ptr if a is not None else None
        raw = lambda a: None if a is None else getattr(a, "raw_ptr", None) or a.ptr
        _lib.check(self._lib.dfhip_mlp2_ce_backward(x.ptr, h.ptr, dz1.ptr, dlogits.ptr, dw1.ptr, g(db1), dw2.ptr, g(db2),
                                                    g(rowloss), raw(loss), M, K, HD, P))
        if loss is not None and hasattr(loss, "settled"):
            loss.settled()

    def dropout_apply(self, x, mask, y, n, keep, backward):
        _lib.check(self._lib.dfhip_dropout_apply(x.ptr, mask.ptr, y.ptr, int(n), float(keep), 1 if backward else 0))

    def deferred_array(self, size):
        return DeferredHipArray(size, self._lib)

    def loss_finish(self, rowloss, loss, N):
        _lib.check(self._lib.dfhip_loss_finish(rowloss.ptr, getattr(loss, "raw_ptr", None) or loss.ptr, N))

    def softmax_ce_backward(self, logits, target, gloss, dlogits, rows, cols):
        _lib.check(self._lib.dfhip_softmax_ce_backward(logits.ptr, target.ptr, gloss.ptr, dlogits.ptr,
                                                       rows, cols))

    def reduce_to_shape(self, src, dst, src_shape, dst_shape):
        ns, ss = _dims(src_shape)
        nd, ds = _dims(dst_shape)
        _lib.check(self._lib.dfhip_reduce_to_shape(src.ptr, dst.ptr, ns, ss, nd, ds))

    def reduce_axis(self, src, dst, outer, red, inner, op="sum", scale=1.0):
        _lib.check(self._lib.dfhip_reduce_axis(src.ptr, dst.ptr, int(outer), int(red), int(inner),
                                               0 if op == "sum" else 1, float(scale)))

    def broadcast_axis(self, src, dst, outer, red, inner, scale=1.0):
        _lib.check(self._lib.dfhip_broadcast_axis(src.ptr, dst.ptr, int(outer), int(red), int(inner),
                                                  float(scale)))

    def axpy(self, dst, src, alpha=1.0):
        _lib.check(self._lib.dfhip_axpy(dst.ptr, src.ptr, float(alpha), dst.size))

    def copy(self, src, dst):
        _lib.check(self._lib.dfhip_d2d(src.ptr, dst.ptr, dst.size))

    def adam_step(self, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step,
                  grad_scale=1.0):
        n = len(params)
        sizes = (c_int64 * n)(*[p.size for p in params])
        _lib.check(self._lib.dfhip_adam_step(n, _ptr_table(params), _ptr_table(grads), _ptr_table(exp_avg),
                                             _ptr_table(exp_avg_sq), sizes, lr, beta1, beta2, eps,
                                             weight_decay, int(step), grad_scale))

    def adam_hyper(self, hyper, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
        """Write the scalars of Adam step `step` into the device array `hyper` (>= 16 floats)."""
        _lib.check(self._lib.dfhip_adam_hyper(hyper.ptr, lr, beta1, beta2, eps, weight_decay, int(step),
                                              grad_scale))

    def adam_hyper_table(self, hyper, cap, lr, beta1, beta2, eps, weight_decay, first_step, grad_scale=1.0):
        """Fill `hyper` (16 + 16 * cap floats) with the scalars of Adam steps first_step ... first_step + cap - 1."""
        _lib.check(self._lib.dfhip_adam_hyper_table(hyper.ptr, int(cap), lr, beta1, beta2, eps, weight_decay,
                                                    int(first_step), grad_scale))

    def adam_hyper_index(self, hyper):
        """(entries of the table the device has consumed, length of the table); drains the stream."""
        idx, cap = c_int(0), c_int(0)
        _lib.check(self._lib.dfhip_adam_hyper_index(hyper.ptr, byref(idx), byref(cap)))
        return idx.value, cap.value

    def adam_step_dev(self, params, grads, exp_avg, exp_avg_sq, hyper):
        n = len(params)
        sizes = (c_int64 * n)(*[p.size for p in params])
        _lib.check(self._lib.dfhip_adam_step_dev(n, _ptr_table(params), _ptr_table(grads), _ptr_table(exp_avg),
                                                 _ptr_table(exp_avg_sq), sizes, hyper.ptr))

    def sgd_step(self, params, grads, momentum_buf, lr, momentum, weight_decay, nesterov, grad_scale=1.0):
        n = len(params)
        sizes = (c_int64 * n)(*[p.size for p in params])
        mb = _ptr_table(momentum_buf) if momentum_buf else None
        _lib.check(self._lib.dfhip_sgd_step(n, _ptr_table(params), _ptr_table(grads), mb, sizes, lr,
                                            momentum, weight_decay, int(bool(nesterov)), grad_scale))

    def sgd_hyper(self, hyper, lr, momentum, weight_decay, nesterov, grad_scale=1.0):
        _lib.check(self._lib.dfhip_sgd_hyper(hyper.ptr, lr, momentum, weight_decay, int(bool(nesterov)), grad_scale))

    def sgd_step_dev(self, params, grads, momentum_buf, hyper):
        n = len(params)
        sizes = (c_int64 * n)(*[p.size for p in params])
        mb = _ptr_table(momentum_buf) if momentum_buf else None
        _lib.check(self._lib.dfhip_sgd_step_dev(n, _ptr_table(params), _ptr_table(grads), mb, sizes, hyper.ptr))

    def pack(self, srcs, flat):
        n = len(srcs)
        sizes = (c_int64 * n)(*[s.size for s in srcs])
        _lib.check(self._lib.dfhip_pack(n, _ptr_table(srcs), sizes, flat.ptr))

    def unpack(self, dsts, flat):
        n = len(dsts)
        sizes = (c_int64 * n)(*[d.size for d in dsts])
        _lib.check(self._lib.dfhip_unpack(n, _ptr_table(dsts), sizes, flat.ptr))

    # ---- collectives (RCCL over xGMI) ---------------------------------------------------------
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        _lib.check(self._lib.dfhip_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        _lib.check(self._lib.dfhip_comm_init(int(rank), int(world), unique_id))

    def allreduce_sum(self, handle, count=None):
        _lib.check(self._lib.dfhip_allreduce_sum(handle.ptr, handle.size if count is None else int(count)))

    def allreduce_sum_async(self, handle, count=None):
        _lib.check(self._lib.dfhip_allreduce_sum_async(handle.ptr, handle.size if count is None else int(count)))

    def comm_wait(self):
        _lib.check(self._lib.dfhip_comm_wait())

    def comm_check(self):
        """raises when RCCL reports an asynchronous error on the communicator (a rank died, a link failed)"""
        _lib.check(self._lib.dfhip_comm_check())

    def comm_allreduce_max(self, value):
        v = c_float(float(value))
        _lib.check(self._lib.dfhip_comm_allreduce_max(byref(v)))
        return float(v.value)

    def comm_rccl_path(self):
        buf = ctypes.create_string_buffer(512)
        _lib.check(self._lib.dfhip_comm_rccl_path(buf, 512))
        return buf.value.decode()

    def comm_count(self):
        """(ranks, rank) as the RCCL communicator reports them."""
        n, r = c_int(0), c_int(0)
        _lib.check(self._lib.dfhip_comm_count(byref(n), byref(r)))
        return int(n.value), int(r.value)

    def comm_destroy(self):
        self._lib.dfhip_comm_destroy()

    # ---- stream, graphs, timing ---------------------------------------------------------------
    def synchronize(self):
        _lib.check(self._lib.dfhip_sync())

    capturing = False      # a stream capture is open (arrays allocated now belong to that graph)
    capture_feeds = None   # list the capturing front end collects host-drawn graph inputs in (Dropout masks)

    def capture_begin(self):
        _lib.check(self._lib.dfhip_graph_begin())
        self.capturing = True

    def capture_end(self, keep=None):
        h = c_void_p()
        self.capturing = False
        _lib.check(self._lib.dfhip_graph_end(byref(h)))
        return HipGraph(self, h, keep)

    def event(self):
        e = c_void_p()
        _lib.check(self._lib.dfhip_event_create(byref(e)))
        return e

    def event_record(self, e):
        _lib.check(self._lib.dfhip_event_record(e))

    def event_elapsed_ms(self, e0, e1):
        ms = c_float()
        _lib.check(self._lib.dfhip_event_elapsed_ms(e0, e1, byref(ms)))
        return ms.value

    def clock_probe(self, milliseconds=100.0):
        """(MHz the chip holds under a dense fp32-MFMA load, TFLOP/s that load delivered)."""
        mhz, tf = c_double(0.0), c_double(0.0)
        _lib.check(self._lib.dfhip_clock_probe(float(milliseconds), byref(mhz), byref(tf)))
        return float(mhz.value), float(tf.value)


    def pool_stats(self):
        a, b, c = c_int64(), c_int64(), c_int64()
        self._lib.dfhip_pool_stats(byref(a), byref(b), byref(c))
        return {"reserved_bytes": a.value, "in_use_bytes": b.value, "hipmalloc_calls": c.value}

    def prof_enable(self, on=True):
        self._lib.dfhip_prof_enable(int(bool(on)))

    def prof_reset(self):
        self._lib.dfhip_prof_reset()

    def prof_gate_close(self, timeout_ms=100):
        _lib.check(self._lib.dfhip_prof_gate_close(int(timeout_ms)))

    def prof_gate_open(self):
        self._lib.dfhip_prof_gate_open()

    def prof_read(self):
        cap = 64
        names = (c_char_p * cap)()
        ms = (c_double * cap)()
        calls = (c_int64 * cap)()
        flops = (c_double * cap)()
        nbytes = (c_double * cap)()
        kms = (c_double * cap)()
        klaunches = (c_int64 * cap)()
        n = self._lib.dfhip_prof_read(cap, names, ms, calls, flops, nbytes, kms, klaunches)
        return {names[i].decode(): {"ms": ms[i], "calls": calls[i], "flops": flops[i], "bytes": nbytes[i],
                                    "kernel_ms": kms[i], "kernel_launches": klaunches[i]}
                for i in range(n)}


    def prof_read_kernels(self):
        """Per kernel instantiation: {name: {family, launches, kernel_ms, flops, bytes}} (dfhip_prof_read_kernels)."""
        cap = 256
        names, fams = (c_char_p * cap)(), (c_char_p * cap)()
        ms, launches = (c_double * cap)(), (c_int64 * cap)()
        flops, nbytes = (c_double * cap)(), (c_double * cap)()
        n = self._lib.dfhip_prof_read_kernels(cap, names, fams, ms, launches, flops, nbytes)
        out = {}
        for i in range(n):
            key = names[i].decode()
            ent = out.setdefault(key, {"family": fams[i].decode(), "launches": 0, "kernel_ms": 0.0, "flops": 0.0, "bytes": 0.0})
            ent["launches"] += launches[i]
            ent["kernel_ms"] += ms[i]
            ent["flops"] += flops[i]
            ent["bytes"] += nbytes[i]
        return out


_module = None


def get_module():
    """Cached singleton (the reference re-instantiates every device on each Device() call,
    backend_tensor.py:189-196; device factories must therefore be cheap)."""
    global _module
    if _module is None:
        _module = HipModule()
    return _module


unavailable_reason = ""      # why available() last said no (Device('hip') puts it into its error message)


def available():
    """True iff libdfhip.so is built AND a HIP device is visible. Never raises; `unavailable_reason` says why not."""
    global unavailable_reason
    if not _lib.library_present():
        unavailable_reason = f"{_lib.LIB_PATH} is not built"
        return False
    try:
        lib = _lib.load()
        n = c_int()
        rc = lib.dfhip_device_count(byref(n))
        if n.value > 0:
            unavailable_reason = ""
            return True
        unavailable_reason = "no HIP device is visible" + (f" ({lib.dfhip_last_error().decode('utf-8', 'replace')})" if rc else "")
        return False
    except Exception as e:        # e.g. a stale libdfhip.so that lacks a symbol the binding declares
        unavailable_reason = f"{_lib.LIB_PATH} does not load: {type(e).__name__}: {e}"
        return False
