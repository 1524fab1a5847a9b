"""Define-by-run autograd over BackendTensor — host-side mirror of DeepFlows/tensor.py
(Graph :9-53, Tensor :59-537, Unary/BinaryOperator :545-619, op classes :622-974,
factories :1068-1112).

Same public surface and the same tape semantics: `Graph.node_list` is a global
creation-ordered tape, `backward()` walks it in reverse calling `node.grad_fn(parent, grad)`,
leaves keep their gradients, `free_graph()` keeps only leaves.  Two things change when the
device module is a fused one (`device.mod.fused`, i.e. `hip`):
  * gradients that must be summed back to a broadcast parent's shape are reduced ON THE
    DEVICE (`reduce_to_shape`) instead of the reference's D2H -> np.sum -> H2D round trip
    (tensor.py:462-483; 103 round trips per ResNet-18 step);
  * no zero-filled `.grad` is allocated per autograd node at construction (tensor.py:134);
    the first contribution is assigned, later ones are added — same values.
"""
from typing import Optional

import numpy as np

from .autograd import is_grad_enable, no_grad
from .backend_selection import BackendTensor, Device, backend_api, default_device


class Graph:
    """Global tape of the nodes that require grad (tensor.py:9-53)."""
    node_list: list = []

    @classmethod
    def add(cls, node):
        cls.node_list.append(node)

    @classmethod
    def clear(cls):
        cls.node_list.clear()

    @classmethod
    def free_graph(cls):
        survivors = []
        for node in cls.node_list:
            leaf = node.is_leaf          # decide before the edges are cut
            node.children.clear()
            node.parents.clear()
            if leaf:
                survivors.append(node)
        cls.node_list = survivors

    @classmethod
    def free_graph_all(cls):
        for node in cls.node_list:
            node.children.clear()
            node.parents.clear()
        cls.node_list = []


_tensor_count = 0


def _is_fused(device):
    return bool(getattr(device.mod, "fused", False))


class Tensor:
    def __init__(self, array, dtype="float32", device: Optional[Device] = None,
                 name: Optional[str] = None, requires_grad: bool = False) -> None:
        global _tensor_count
        _tensor_count += 1
        self.unique_id = _tensor_count
        self.name = name if name is not None else str(self.unique_id)

        if isinstance(array, Tensor):
            device = array.device if device is None else device
            dtype = array.dtype if dtype is None else dtype
            if device == array.device and dtype == array.dtype:
                self.data = array.data
            else:
                self.data = Tensor._from_numpy(array.numpy(), device=device, dtype=dtype)
        elif isinstance(array, BackendTensor):
            self.data = array
        else:
            device = device if device else default_device()
            self.data = Tensor._from_numpy(array, device=device, dtype=dtype)

        self.requires_grad = bool(requires_grad and is_grad_enable())
        if self.requires_grad and not getattr(self.data.device.mod, "lazy_grad", False):
            self.grad = backend_api.zeros_like(self.data)     # tensor.py:134
        else:
            self.grad = None
        self.children = []
        self.parents = []
        if self.requires_grad:
            Graph.add(self)

    @staticmethod
    def _from_numpy(numpy_array, device, dtype):
        return backend_api.Btensor(np.asarray(numpy_array), dtype=dtype, device=device)

    @staticmethod
    def make_const(source, requires_grad=False):
        t = Tensor.__new__(Tensor)
        t.__init__(source, requires_grad=False)
        return t

    @property
    def is_leaf(self):
        return self.requires_grad and len(self.parents) == 0

    shape = property(lambda self: self.data.shape)
    ndim = property(lambda self: self.data.ndim)
    dtype = property(lambda self: self.data.dtype)
    size = property(lambda self: self.data.size)
    device = property(lambda self: self.data.device)

    def numpy(self):
        return self.data.numpy()

    def detach(self):
        return Tensor.make_const(self)

    def dispose(self):
        if self.grad is not None and not self.is_leaf:
            self.grad = None
        self.children.clear()
        self.parents.clear()
        try:
            Graph.node_list.remove(self)
        except ValueError:
            pass

    @property
    def T(self):
        return self.transpose()

    def reshape(self, *new_shape):
        return Reshape(self, new_shape)

    def transpose(self, *axes):
        return transpose(self, axes if len(axes) != 0 else None)

    def max(self, axis=None, keepdims=False):
        return max(self, axis, keepdims)

    def sum(self, axis=None, keepdims=False):
        return sum(self, axis, keepdims)

    def mean(self, axis=None, keepdims=False):
        return mean(self, axis, keepdims)

    def build_edge(self, node):
        # A tensor that does not require grad is never on the tape, so free_graph() never clears its
        # children: a long-lived input (the fixed batch of a benchmark, a captured step's input)
        # would keep every step's first node — and its activations — alive.  It gets no child link;
        # the node still lists it as a parent (backward skips parents that do not require grad).
        if self.requires_grad:
            self.children.append(node)
        node.parents.append(self)

    # operator overloads
    def __add__(self, x): return add(self, x)
    def __radd__(self, x): return add(x, self)
    def __sub__(self, x): return sub(self, x)
    def __rsub__(self, x): return sub(x, self)
    def __mul__(self, x): return mul(self, x)
    def __rmul__(self, x): return mul(x, self)
    def __matmul__(self, x): return matmul(self, x)
    def __rmatmul__(self, x): return matmul(x, self)
    def __truediv__(self, x): return div(self, x)
    def __rtruediv__(self, x): return div(x, self)
    def __pow__(self, x): return pow(self, x)
    def __rpow__(self, x): return pow(x, self)
    def __len__(self): return len(self.data)
    def __pos__(self): return self * 1
    def __neg__(self): return self * -1
    def __getitem__(self, key): return get_slice(self, key)

    def __setitem__(self, key, value):
        assert not self.requires_grad, "In-place operation is forbidden in node requires grad."
        if isinstance(key, Tensor):
            key = key.data
        self.data[key] = value.data if isinstance(value, Tensor) else value

    def _inplace(self, other, fn):
        assert not self.requires_grad, "In-place operation is forbidden in node requires grad."
        other = other.data if isinstance(other, Tensor) else other
        self.data = fn(self.data, other)
        return self

    def __iadd__(self, o): return self._inplace(o, lambda a, b: a + b)
    def __isub__(self, o): return self._inplace(o, lambda a, b: a - b)
    def __imul__(self, o): return self._inplace(o, lambda a, b: a * b)
    def __itruediv__(self, o): return self._inplace(o, lambda a, b: a / b)
    def __imatmul__(self, o): return self._inplace(o, lambda a, b: a @ b)

    def _compare(self, other, fn):
        with no_grad():
            other = other.data if isinstance(other, Tensor) else other
            return Tensor(fn(self.data, other))

    def __lt__(self, o): return self._compare(o, lambda a, b: a < b)
    def __le__(self, o): return self._compare(o, lambda a, b: a <= b)
    def eq(self, o): return self._compare(o, lambda a, b: a == b)
    def ne(self, o): return self._compare(o, lambda a, b: a != b)
    def __gt__(self, o): return self._compare(o, lambda a, b: a > b)
    def __ge__(self, o): return self._compare(o, lambda a, b: a >= b)

    # ---- reverse sweep (tensor.py:421-499) ---------------------------------------------------
    def backward(self, retain_graph: bool = False, milestone=None):
        """milestone = (node, callback) (extension): callback() runs when the reverse sweep reaches
        `node`, i.e. when every node created after it has been processed — the gradients of all
        parameters used only behind it are final (data-parallel overlap, graph_step.py)."""
        if self not in Graph.node_list:
            return
        if self.data.ndim != 1:
            raise ValueError("backward should be called only on a scalar.")
        # a device may queue the weight gradients of the sweep and run them as one launch at its
        # end (hip: dfhip_wgrad_group_begin/_end); nothing reads a parameter's .grad before that
        group_begin = getattr(self.device.mod, "wgrad_group_begin", None)
        grouping = bool(group_begin()) if group_begin is not None else False
        try:
            self._backward_sweep(milestone)
        finally:
            if grouping:
                self.device.mod.wgrad_group_end()
        if not retain_graph:
            Graph.free_graph()

    def _backward_sweep(self, milestone):
        with no_grad():
            self.grad = _grad_seed(self.data)
            start = len(Graph.node_list) - 1
            while Graph.node_list[start] is not self:
                start -= 1
            for node in Graph.node_list[start::-1]:
                if milestone is not None and node is milestone[0]:
                    milestone[1]()
                    milestone = None
                grad = node.grad
                if grad is None:
                    continue
                if isinstance(grad, Tensor):
                    grad = grad.data
                for parent in [p for p in node.parents if p.requires_grad]:
                    held = parent.grad
                    if (held is not None and getattr(node, "fuses_accumulation", False)
                            and node.x is parent and hasattr(held.device.mod, "conv2d_dgrad_add")):
                        # a fused node adds the gradient the parent already holds in its own kernel
                        res = node.grad_fn(parent, grad, held)
                        if res is not None:
                            parent.grad = res
                        continue
                    contrib = node.grad_fn(parent, grad)
                    if contrib is None:          # (a fused sibling node already delivered this contribution)
                        continue
                    if isinstance(contrib, Tensor):
                        contrib = contrib.data
                    if tuple(contrib.shape) != tuple(parent.shape):
                        contrib = _unbroadcast(contrib, parent.shape, self.device)
                    parent.grad = contrib if parent.grad is None else parent.grad + contrib
                if not node.is_leaf:
                    node.grad = None

    def zero_grad(self):
        self.grad = None

    def to(self, device):
        if isinstance(device, str):
            device = backend_api.Device(device)
        if self.device == device:
            return self
        return Tensor(self.data, dtype=self.dtype, device=device)

    def cpu(self):
        return self.to("cpu")

    def cuda(self):
        return self.to("cuda")

    def hip(self):
        return self.to("hip")

    def __repr__(self):
        return f"Tensor({self.data}, requires_grad={self.requires_grad}, device={self.device})"

    def __str__(self):
        return str(self.data)


_SEEDS = {}


def _grad_seed(data):
    """ones_like(loss) (tensor.py:428).  Gradients are never modified in place, so the (1,) seed of a
    device is built once and shared by every backward() -- one fill launch less per step.  Not cached
    while a stream capture is open (the array would belong to that graph)."""
    if tuple(int(v) for v in data.shape) != (1,):
        return backend_api.ones_like(data)
    mod = data.device.mod
    if not hasattr(mod, "capturing"):            # (devices without captures -- the test oracle -- are not cached)
        return backend_api.ones_like(data)
    seed = _SEEDS.get(id(mod))
    if seed is None:
        seed = backend_api.ones_like(data)
        # a node that already knows its gradient for an upstream value of exactly 1 (cross-entropy on the hip
        # device) recognises the seed by this mark and launches nothing
        seed.is_unit_seed = True
        if not mod.capturing:
            _SEEDS[id(mod)] = seed
    return seed


def _unbroadcast(grad, shape, loss_device):
    """Sum `grad` back to `shape` (the parent was broadcast in the forward)."""
    shape = tuple(int(s) for s in shape)
    fast = getattr(grad.device.mod, "reduce_to_shape", None)
    if fast is not None:
        g = grad.compact()
        out = BackendTensor.make(shape, device=grad.device)
        fast(g._handle, out._handle, g.shape, shape)
        return out
    # reference path: through host NumPy (tensor.py:462-483)
    g = grad.numpy()
    if g.ndim > len(shape):
        g = np.sum(g, axis=tuple(range(g.ndim - len(shape))))
    axes = tuple(i for i, (a, b) in enumerate(zip(g.shape, shape)) if a != b)
    if axes:
        g = np.sum(g, axis=axes, keepdims=True)
    return backend_api.Btensor(g, device=loss_device)


# ---- deferred results (fused devices) ------------------------------------------------------------------
class LazyTensor(Tensor):
    """The result of a module call that has NOT been launched yet (nn/lazy.py decides when a call is deferred).

    The reference's model scripts are sequences of plain module calls -- `out = self.conv1(x); out = self.bn1(out); out =
    self.conv2(out); out = self.bn2(out); out = out + identity` (test/ResNet_CIFAR10_cuda.py:31-47) -- while the hand-written
    kernels of the hip device cover several of those calls at once (BatchNorm applied on the next convolution's operand
    loads, the block's two BatchNorms and its sum in one pass, ...).  A deferred result records the call; the NEXT call
    either extends the record (and stays deferred) or is the point where the fused entry point for the recorded sequence
    is known and launched.  Anything else that touches the object -- an attribute, an operator, a node constructor --
    forces it: the recorded call runs as the ordinary node it always was, and the object forwards to that node from then
    on.  It is a Tensor for `isinstance`, never sits on the tape, and is never a parent of a node (constructors take
    `realize(x)`)."""

    def __init__(self, op, x, module=None, extra=None):     # (no Tensor.__init__: nothing exists yet)
        object.__setattr__(self, "_lz", [op, x, module, extra, None])   # [op, input, module, extra, forced node]

    def __getattribute__(self, name):
        if name == "_lz" or name == "__class__":
            return object.__getattribute__(self, name)
        return getattr(realize(self), name)

    def __setattr__(self, name, value):
        setattr(realize(self), name, value)

    # (operators are looked up on the type: the inherited ones build nodes, whose constructors realize their inputs;
    # `+` is the tail of a residual block)
    def __add__(self, other):
        return _lazy_hooks["add"](self, other)

    def __repr__(self):
        return repr(realize(self))

    def __str__(self):
        return str(realize(self))

    def __len__(self):
        return len(realize(self))


_lazy_hooks = {}          # filled by nn/lazy.py: "force" (a deferred call -> its node), "add", "mean"


def realize(x):
    """x itself, or the node a deferred result stands for (launching its recorded call now)"""
    if type(x) is LazyTensor:
        st = object.__getattribute__(x, "_lz")
        if st[4] is None:
            st[4] = _lazy_hooks["force"](x)
        return st[4]
    return x


# ---- operator bases ------------------------------------------------------------------------
class UnaryOperator(Tensor):
    def __init__(self, x: Tensor) -> None:
        x = realize(x)
        if not isinstance(x, Tensor):
            x = Tensor(x)
        super().__init__(array=self.forward(x), device=x.device,
                         requires_grad=is_grad_enable() and x.requires_grad)
        if self.requires_grad:
            x.build_edge(self)

    def forward(self, x):
        raise NotImplementedError

    def grad_fn(self, x, grad):
        raise NotImplementedError

    def __repr__(self):
        return f"Tensor({self.data}, op={self.__class__.__name__})"


class BinaryOperator(Tensor):
    def __init__(self, x, y) -> None:
        x, y = realize(x), realize(y)
        if not isinstance(x, Tensor):
            # scalar / ndarray on the left: lift it to the device of the tensor operand
            x = Tensor(np.asarray(x, dtype=np.float32).reshape(-1) if np.ndim(x) == 0 else x,
                       device=y.device)
        if isinstance(y, BackendTensor):
            assert x.device == y.device
            fx, fy = x.data, y
            y = Tensor(y)
        elif isinstance(y, Tensor):
            assert x.device == y.device
            fx, fy = x.data, y.data
        else:
            fx, fy = x.data, np.float32(y)
            if _is_fused(x.device):
                y = _Scalar(fy)                      # no 1-element H2D per constant
            else:
                y = Tensor(np.array([y]), device=x.device)   # tensor.py:597-599
        super().__init__(array=self.forward(fx, fy), device=x.device,
                         requires_grad=is_grad_enable() and (x.requires_grad or y.requires_grad))
        if self.requires_grad:
            x.build_edge(self)
            y.build_edge(self)

    def forward(self, x, y):
        raise NotImplementedError

    def __repr__(self):
        return f"Tensor({self.data}, op={self.__class__.__name__})"


class _Scalar:
    """Constant right operand kept on the host (fused devices only)."""
    requires_grad = False

    def __init__(self, value):
        self.data = value
        self.children = []
        self.parents = []
        self.ndim = 0

    def build_edge(self, node):
        node.parents.append(self)


class add(BinaryOperator):
    def forward(self, x, y):
        return x + y

    def grad_fn(self, node, grad):
        return grad


class sub(BinaryOperator):
    def forward(self, x, y):
        return x - y

    def grad_fn(self, node, grad):
        return grad if node is self.parents[0] else -grad


class mul(BinaryOperator):
    def forward(self, x, y):
        return x * y

    def grad_fn(self, node, grad):
        other = self.parents[1] if node is self.parents[0] else self.parents[0]
        return grad * other.data


class div(BinaryOperator):
    def forward(self, x, y):
        return x / y

    def grad_fn(self, node, grad):
        q = grad / self.parents[1].data
        return q if node is self.parents[0] else -self.data * q


class pow(BinaryOperator):
    def forward(self, x, y):
        return x ** y

    def grad_fn(self, node, grad):
        if node is self.parents[0]:
            return (self.data * self.parents[1].data / node.data) * grad
        return self.data * backend_api.log(self.parents[0].data) * grad


class matmul(BinaryOperator):
    def forward(self, x, y):
        return x @ y

    def grad_fn(self, node, grad):
        a, b = self.parents
        if node is a:
            if b.ndim == 1:
                return backend_api.expand_dims(grad, -1) @ backend_api.expand_dims(b.data, -2)
            return grad @ b.data.transpose()       # dA = dC . B^T (tensor.py:708)
        if a.ndim == 1:
            return backend_api.expand_dims(a.data, -1) @ backend_api.expand_dims(grad, -2)
        return a.data.transpose() @ grad           # dB = A^T . dC (tensor.py:716)


class sum(UnaryOperator):
    def __init__(self, x, axes=None, keepdims=False) -> None:
        self.axes = axes
        self.keepdims = keepdims
        super().__init__(x)

    def forward(self, x):
        a = x.data
        if isinstance(self.axes, (list, tuple)) and len(self.axes) > 1:
            for axis in reversed(sorted(self.axes)):
                a = a.sum(axis=axis)
            return a
        return backend_api.summation(a, axis=self.axes, keepdims=self.keepdims)

    def grad_fn(self, x, grad):
        # only keepdims=True / axis=None are safe, as in the reference (tensor.py:747-750)
        return backend_api.ones(x.shape, device=x.device) * grad


class mean(UnaryOperator):
    def __new__(cls, x, axis=None, keepdims=False):
        # (two `mean(axis=2)` calls and a Linear are one kernel on the hip device: the call may be deferred, nn/lazy.py)
        hook = _lazy_hooks.get("mean")
        lz = hook(x, axis, keepdims) if hook is not None else None
        return lz if lz is not None else super().__new__(cls)

    def __init__(self, x, axis=None, keepdims=False) -> None:
        self.axis = axis
        self.keepdims = keepdims
        super().__init__(x)

    def forward(self, x):
        return backend_api.mean(x.data, axis=self.axis, keepdims=self.keepdims)

    def grad_fn(self, x, grad):
        scale = self.data.size / x.data.size
        fast = getattr(x.device.mod, "broadcast_axis", None)
        if fast is not None and self.axis is not None and not isinstance(self.axis, (tuple, list)):
            ax = self.axis
            out = BackendTensor.make(x.shape, device=x.device)
            g = grad.compact()
            shp = x.shape
            fast(g._handle, out._handle, _prod(shp[:ax]), shp[ax], _prod(shp[ax + 1:]), scale)
            return out
        if not (self.axis is None or self.keepdims):
            grad = backend_api.expand_dims(grad, axis=self.axis)
        return backend_api.ones(x.shape, device=x.device) * grad * self.data.size / x.data.size


def _prod(xs):
    out = 1
    for v in xs:
        out *= int(v)
    return out


class max(UnaryOperator):
    def __init__(self, x, axis=None, keepdims=False) -> None:
        self.axis = axis
        self.keepdims = keepdims
        super().__init__(x)

    def forward(self, x):
        return backend_api.max(x.data, axis=self.axis, keepdims=self.keepdims)

    def grad_fn(self, x, grad):
        if self.keepdims or self.axis is None:
            full = self.data
        else:
            full = backend_api.expand_dims(self.data, axis=self.axis)
            grad = backend_api.expand_dims(grad, axis=self.axis)
        full = full.broadcast_to(x.data.shape)
        return (full == x.data) * grad            # every tied maximum gets the gradient


class exp(UnaryOperator):
    def forward(self, x):
        return backend_api.exp(x.data)

    def grad_fn(self, x, grad):
        return self.data * grad


class log(UnaryOperator):
    def forward(self, x):
        return backend_api.log(x.data)

    def grad_fn(self, x, grad):
        return grad / x.data


class maximum(BinaryOperator):
    def __new__(cls, x, y=None):
        # maximum(x, 0) IS the reference's ReLU (nn/functional.py:15-16 is this very call; test/MobileNet.py:34 writes it out):
        # on a fused device the one-launch node of F.relu instead of three elementwise launches (nn/lazy.py)
        hook = _lazy_hooks.get("maximum0")
        if hook is not None and isinstance(y, (int, float)) and not isinstance(y, bool) and y == 0:
            out = hook(x)
            if out is not None:
                return out
        return super().__new__(cls)

    def forward(self, x, y):
        return backend_api.maximum(x, y)

    def grad_fn(self, x, grad):
        return (self.data == x.data) * grad


def sqrt(x):
    return x ** 0.5


def square(x):
    return x * x


class Reshape(UnaryOperator):
    def __init__(self, x, new_shape):
        self.new_shape = new_shape
        super().__init__(x)

    def forward(self, x):
        return x.data.compact().reshape(self.new_shape)

    def grad_fn(self, x, grad):
        return grad.compact().reshape(x.shape)


class transpose(UnaryOperator):
    def __init__(self, x, axes=None):
        self.axes = axes
        super().__init__(x)

    def forward(self, x):
        return x.data.transpose(self.axes)

    def grad_fn(self, x, grad):
        if self.axes is None:
            return grad.transpose()
        return grad.transpose(tuple(int(a) for a in np.argsort(self.axes)))


class get_slice(UnaryOperator):
    def __init__(self, x, key):
        self.key = key.data if isinstance(key, Tensor) else key
        super().__init__(x)

    def forward(self, x):
        return x.data[self.key]

    def grad_fn(self, x, grad):
        full = backend_api.zeros(x.shape, device=x.device)
        full[self.key] = grad
        return full


# ---- factories (tensor.py:1068-1112) ----------------------------------------------------------
def empty(shape, dtype=None, device=None, requires_grad=False):
    return Tensor(np.empty(shape), dtype=dtype, device=device, requires_grad=requires_grad)


def zeros(shape, dtype=None, device=None, requires_grad=False):
    return Tensor(np.zeros(shape), dtype=dtype, device=device, requires_grad=requires_grad)


def ones(shape, dtype=None, device=None, requires_grad=False):
    return Tensor(np.ones(shape), dtype=dtype, device=device, requires_grad=requires_grad)


def randn(*shape, dtype=None, device=None, requires_grad=False):
    return Tensor(np.random.randn(*shape), dtype=dtype, device=device, requires_grad=requires_grad)


def rand(*shape, dtype=None, device=None, requires_grad=False):
    return Tensor(np.random.rand(*shape), dtype=dtype, device=device, requires_grad=requires_grad)


def uniform(low, high, shape=None, dtype=None, device=None, requires_grad=False):
    return Tensor(np.random.uniform(low, high, size=shape), dtype=dtype, device=device,
                  requires_grad=requires_grad)
