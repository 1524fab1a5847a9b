"""nn.functional — host-side mirror of DeepFlows/nn/functional.py.

Every function keeps the reference's name and argument meaning.  Each has two bodies:
  * the PRIMITIVE path restates the reference's composition over the 25 device primitives
    (pad -> im2col by k*k strided setitems -> GEMM, softmax-CE as max/exp/sum/log chains, ...);
    it runs on any device module and is what the NumPy oracle and the CPU baseline execute;
  * the FUSED path (taken when `device.mod.fused`) is one autograd node backed by the
    hand-written HIP kernels of libdfhip.so (implicit-GEMM conv, one-pass BN, ...), with
    gradients defined to equal what the reference's autograd produces for the same graph —
    including its quirks: ReLU passes the gradient at x == 0, every tied maximum of a pooling
    window receives the gradient, and the conv input gradient is "last writer wins"
    (functional.py:285-294) unless the device is switched to conv_dgrad_mode="accumulate".
"""
import os
import weakref

import numpy as np

from .. import backend_api, tensor
from ..autograd import is_grad_enable
from ..backend_selection import BackendTensor
from ..tensor import Tensor, realize


def _fused(t):
    return bool(getattr(t.device.mod, "fused", False))


def _new(shape, device):
    return BackendTensor.make(tuple(int(s) for s in shape), device=device)


def _grad_out(param, shape, device):
    """The array a kernel writes `param`'s gradient into: the parameter's slice of the data-parallel gradient bucket
    (`grad_slot`, parallel.GradientBucket) when it has one and this is the first contribution of the sweep -- the all-reduce
    then runs on what the backward kernels wrote, no pack launch --, a new array otherwise."""
    slot = getattr(param, "grad_slot", None)
    if slot is not None and param.grad is None and tuple(slot.shape) == tuple(int(s) for s in shape):
        return slot
    return _new(shape, device)


# rows of the BatchNorm partials buffer a fused convolution may fill (one per workgroup)
_BN_PARTIAL_ROWS = 512

# ---- BatchNorm2d backward without a launch of its own (include/dfhip.h: dfhip_bn_bwd_in / dfhip_bn_stats_out) -----------
# The autograd chain of nn/modules/batchnorm.py:30-55 needs sum(dy) and sum(dy * (z - mean)) over the whole batch before
# it can form dz.  On a fused device the kernel that PRODUCES dy (the input gradient of the next convolution) leaves those
# sums behind as partial rows (tag `bn_rows` on the gradient it returns), the BatchNorm node then hands its parent -- the
# convolution whose output z is -- the SAME dy tagged with a _PendingBN (tag `bn_pending`), and that convolution's input- and
# weight-gradient kernels form dz on their operand loads.  Anything that cannot take a pending gradient writes it out first
# (_pending_materialise); DF_BN_BWD_FUSED=0 switches the whole mechanism off.
_BN_BWD_FUSED = os.environ.get("DF_BN_BWD_FUSED", "1") != "0"
# (The two convolutions that read a block's input when its shortcut is a convolution do NOT take pending gradients: their one
# launch with two pending operands was measured slower than the two stand-alone BatchNorm launches it replaces -- layer2's
# first block: 9.2 -> 24 us against 5.2 + 6.5 us, round 4.)


class _PendingBN:
    """dz = A dy + B (z - mean) + Cc per channel, not yet applied; handles as the C ABI wants them."""
    __slots__ = ("z", "rows", "nrows", "cap_rows", "K", "kidx", "mean", "var", "gamma", "eps", "m", "dgamma", "dbeta",
                 "table", "table_ready", "geo", "keep")

    def __init__(self, z, rows, nrows, cap_rows, K, kidx, mean, var, gamma, eps, geo, dgamma, dbeta, table, keep):
        self.z, self.rows, self.nrows, self.cap_rows, self.K, self.kidx = z, rows, nrows, cap_rows, K, kidx
        self.mean, self.var, self.gamma, self.eps = mean, var, gamma, eps
        self.geo = geo                      # (N, O, HW) of dy
        self.m = geo[0] * geo[2]
        self.dgamma, self.dbeta, self.table = dgamma, dbeta, table
        self.table_ready = False            # the table / dgamma / dbeta have been written (by an input-gradient launch)
        self.keep = keep                    # the BackendTensors behind the handles


class _BnRows:
    """what a producer left behind for the BatchNorm node `owner`: rows (nrows of a buffer with room for cap_rows, x C x K),
    sums 0 and 1 + j are layer j's"""
    __slots__ = ("_owner", "rows", "nrows", "cap_rows", "K")

    def __init__(self, owner, rows, nrows, cap_rows, K):
        # (a weak reference: the owner's cached gradients hold the tensor that carries this tag -- a strong one would be a
        # reference cycle, and the arrays of a step would wait for the cycle collector instead of going back to the pool)
        self._owner, self.rows, self.nrows, self.cap_rows, self.K = weakref.ref(owner), rows, nrows, cap_rows, K

    @property
    def owner(self):
        return self._owner()


def _alias(t):
    """the same storage as another BackendTensor object (tags do not travel)"""
    t = t.compact()
    return BackendTensor.make(t.shape, device=t.device, handle=t._handle)


def _pending_table(pend, dev):
    if not pend.table_ready:
        dev.bn_bwd_table(pend, pend.geo[1])
        pend.table_ready = True


def _pending_materialise(pend, dy):
    """the pending BatchNorm backward written out: a plain gradient tensor (and dgamma / dbeta filled)"""
    dev = dy.device
    _pending_table(pend, dev)
    dz = _new(dy.shape, dev)
    N, O, HW = pend.geo
    dev.bn2d_backward_apply_table(dy.compact()._handle, pend.z, pend.table, dz._handle, N, O, HW)
    return dz


def _plain(grad):
    """`grad` as a gradient tensor without tags (a pending BatchNorm backward is written out)"""
    pend = getattr(grad, "bn_pending", None)
    return grad if pend is None else _pending_materialise(pend, grad)


def _accepts_pending(znode):
    """the convolution node `znode` takes its output gradient as a pending BatchNorm backward"""
    if not _BN_BWD_FUSED or not isinstance(znode, (_Conv2dFused, _BnConv2dFused)) or len(znode.children) != 1:
        return False
    mod = znode.device.mod
    if not hasattr(mod, "conv2d_dgrad_bn") or getattr(mod, "conv_dgrad_mode", "") != "overwrite":
        return False
    return znode._pending_ok()


def _stats_targets(xnode, me, sib=None):
    """[(z handle, mean handle), ...] of the BatchNorm layers whose output gradient the producer `me` is about to COMPLETE
    (every other consumer of `xnode` was created after `me`, so the reverse sweep has been there; `sib`: a node whose
    contribution `me` delivers in the same launch), or None."""
    if not _BN_BWD_FUSED or not isinstance(xnode, (_BatchNorm2dFused, _BatchNorm2dPairFused)):
        return None
    for ch in xnode.children:
        if ch is me or ch is sib:
            continue
        if ch.unique_id < me.unique_id:
            return None
    return xnode._deferred_targets()


def _new_rows(cap, C, K, dev):
    return _new((cap, C, K), dev)


def _tag_rows(out, xnode, rows, nrows, cap, K):
    out.bn_rows = _BnRows(xnode, rows, nrows, cap, K)



class _FusedNode(Tensor):
    """Autograd node produced by one fused kernel; parents are the differentiable inputs."""

    def __init__(self, data, inputs):
        req = is_grad_enable() and any(p is not None and p.requires_grad for p in inputs)
        super().__init__(array=data, device=data.device, requires_grad=req)
        if self.requires_grad:
            for p in inputs:
                if p is not None:
                    p.build_edge(self)

    def __repr__(self):
        return f"Tensor({self.data}, op={self.__class__.__name__})"


# ---- linear / activations ---------------------------------------------------------------------
class _LinearFused(_FusedNode):
    def __init__(self, x, w, b):
        self.x, self.w, self.b = x, w, b
        m, k = x.shape
        p = w.shape[1]
        xd, wd = x.data.compact(), w.data.compact()
        out = _new((m, p), x.device)
        x.device.gemm(xd._handle, wd._handle, b.data.compact()._handle if b is not None else None,
                      out._handle, m, k, p, False, False)
        self._saved = (xd, wd)
        self._grads = None
        super().__init__(out, [x, w, b])

    def grad_fn(self, node, grad):
        xd, wd = self._saved
        g = grad.compact()
        m, k = xd.shape
        p = wd.shape[1]
        dev = g.device
        if hasattr(dev.mod, "linear_backward"):
            # dx, dw and db of the layer as one call (one launch for a classifier-sized layer)
            if self._grads is None or self._grads[0] is not grad:
                dx = _new((m, k), dev) if self.x.requires_grad else None
                dw = _grad_out(self.w, (k, p), dev) if self.w.requires_grad else None
                db = _grad_out(self.b, self.b.shape, dev) if (self.b is not None and self.b.requires_grad) else None
                h = lambda t: t._handle if t is not None else None
                dev.linear_backward(xd._handle, wd._handle, g._handle, h(dx), h(dw), h(db), m, k, p)
                self._grads = (grad, dx, dw, db)
            _, dx, dw, db = self._grads
            return dx if node is self.x else (dw if node is self.w else db)
        if node is self.x:
            out = _new((m, k), dev)
            dev.gemm(g._handle, wd._handle, None, out._handle, m, p, k, False, True)   # dY . W^T
            return out
        if node is self.w:
            out = _grad_out(self.w, (k, p), dev)
            dev.gemm(xd._handle, g._handle, None, out._handle, k, m, p, True, False)   # X^T . dY
            return out
        out = _grad_out(self.b, self.b.shape, dev)
        dev.reduce_to_shape(g._handle, out._handle, g.shape, self.b.shape)
        return out


def linear(input: Tensor, weight: Tensor, bias=None):
    input = realize(input)
    if _fused(input) and input.ndim == 2 and weight.ndim == 2:
        return _LinearFused(input, weight, bias)
    affine = input @ weight
    if bias is not None:
        affine = affine + bias
    return affine


class _ReLUFused(_FusedNode):
    def __init__(self, x):
        self.xd = x.data.compact()
        out = _new(x.shape, x.device)
        x.device.relu_forward(self.xd._handle, out._handle)
        super().__init__(out, [x])

    def grad_fn(self, node, grad):
        out = _new(self.xd.shape, grad.device)
        grad.device.relu_backward(self.xd._handle, self.data._handle, grad.compact()._handle, out._handle)
        return out


class _DropoutFused(_FusedNode):
    """x * mask / (1 - p) (nn/modules/dropout.py:31) as one node: one launch each way with the two primitive nodes' values bit
    for bit (dfhip_dropout_apply); the mask is the host-drawn array the module uploaded"""

    def __init__(self, x, mask, keep):
        self.mask, self.keep = mask, float(keep)
        xd = x.data.compact()
        out = _new(x.shape, x.device)
        x.device.mod.dropout_apply(xd._handle, mask._handle, out._handle, xd._handle.size, self.keep, False)
        super().__init__(out, [x])

    def grad_fn(self, node, grad):
        g = grad.compact()
        out = _new(g.shape, g.device)
        g.device.mod.dropout_apply(g._handle, self.mask._handle, out._handle, g._handle.size, self.keep, True)
        return out


def dropout_train(x: Tensor, mask, keep):
    """x * mask / keep for a mask BackendTensor of x's shape"""
    x = realize(x)
    if _fused(x) and hasattr(x.device.mod, "dropout_apply") and is_grad_enable():
        return _DropoutFused(x, mask.compact(), keep)
    return x * mask / keep


_lazy_relu = None           # nn/lazy.py: relu of a recorded call (relu(fc(x)), relu(bn(conv(x)))) extends or completes the record


def relu(input: Tensor) -> Tensor:
    if _lazy_relu is not None:
        out = _lazy_relu(input)
        if out is not None:
            return out
    input = realize(input)
    if _fused(input):
        return _ReLUFused(input)
    return tensor.maximum(input, 0)


class _LinearReluFused(_FusedNode):
    """relu(linear(x, W, b)) as ONE node: one forward launch (the pre-activation is kept), one backward launch (the
    ReLU's gradient mask -- output == input, tensor.py:872-877 -- is applied on the operand loads of dx / dW / db).
    For the launch-bound layers of the MLP workload (test/MLP_MNIST_cuda.py:44-56): three launches forward and four
    backward otherwise."""

    @classmethod
    def build(cls, x, w, b):
        dev = x.device
        fn = getattr(dev.mod, "linear_relu_forward", None)
        if fn is None or x.ndim != 2 or w.ndim != 2:
            return None
        m, k = (int(v) for v in x.shape)
        p = int(w.shape[1])
        xd, wd = x.data.compact(), w.data.compact()
        z, r = _new((m, p), dev), _new((m, p), dev)
        if not fn(xd._handle, wd._handle, b.data.compact()._handle if b is not None else None, z._handle, r._handle, m, k, p):
            return None
        node = cls.__new__(cls)
        node.x, node.w, node.b = x, w, b
        node._saved = (xd, wd, z)
        node._grads = None
        _FusedNode.__init__(node, r, [x, w, b])
        return node

    def grad_fn(self, node, grad):
        if self._grads is None or self._grads[0] is not grad:
            xd, wd, z = self._saved
            dev = grad.device
            m, k = xd.shape
            p = wd.shape[1]
            dx = _new((m, k), dev) if self.x.requires_grad else None
            dw = _grad_out(self.w, (k, p), dev) if self.w.requires_grad else None
            db = _grad_out(self.b, self.b.shape, dev) if (self.b is not None and self.b.requires_grad) else None
            h = lambda t: t._handle if t is not None else None
            g = grad.compact()
            if not dev.linear_relu_backward(xd._handle, wd._handle, z._handle, self.data._handle, g._handle, h(dx), h(dw),
                                            h(db), m, k, p):
                # (a layer the forward kernel took always has the backward kernel; kept for completeness)
                masked = _new((m, p), dev)
                dev.relu_backward(z._handle, self.data._handle, g._handle, masked._handle)
                dev.linear_backward(xd._handle, wd._handle, masked._handle, h(dx), h(dw), h(db), m, k, p)
            self._grads = (grad, dx, dw, db)
        _, dx, dw, db = self._grads
        return dx if node is self.x else (dw if node is self.w else db)




class _MlpHidden(_LinearReluFused):
    """relu(fc1(x)) of `mlp2_ce_train`: the node _LinearReluFused would be, its arrays filled by the fused forward launch.  When
    the gradient that arrives is the one the fused backward launch was given (d loss / d pre-activation, handed over by
    _MlpLogits), the layer's parameter gradients are the ones that launch formed; any other gradient takes the ordinary
    backward launch."""
    _pre = None                # (dz1 BackendTensor, dW1, db1)

    def grad_fn(self, node, grad):
        if self._pre is not None and grad is self._pre[0]:
            return self._pre[1] if node is self.w else (self._pre[2] if node is self.b else None)
        return super().grad_fn(node, grad)


class _MlpLogits(_LinearFused):
    """fc2(h) of `mlp2_ce_train`.  When the gradient that arrives is the d loss / d logits of the fused forward launch and
    nothing else reads h, ONE launch forms every parameter gradient of both layers (dW1, db1, dW2, db2) and finishes the
    loss (dfhip_mlp2_ce_backward); h's node is then handed d loss / d pre-activation, which it recognises."""

    def grad_fn(self, node, grad):
        hn = self.x
        if self._grads is not None and self._grads[0] is grad:
            _, dx, dw2, db2 = self._grads
            return dx if node is self.x else (dw2 if node is self.w else db2)
        if (grad is self.dl_pre and isinstance(hn, _MlpHidden) and len(hn.children) == 1 and hn.children[0] is self
                and hn.grad is None and not hn.x.requires_grad):
            if self._grads is None or self._grads[0] is not grad:
                dev = grad.device
                xd, w1d, _z = hn._saved
                hd, w2d = self._saved
                M, K = (int(v) for v in xd.shape)
                HD, P = (int(v) for v in w2d.shape)
                dw1 = _grad_out(hn.w, (K, HD), dev)
                db1 = _grad_out(hn.b, hn.b.shape, dev) if (hn.b is not None and hn.b.requires_grad) else None
                dw2 = _grad_out(self.w, (HD, P), dev)
                db2 = _grad_out(self.b, self.b.shape, dev) if (self.b is not None and self.b.requires_grad) else None
                hh = lambda t: t._handle if t is not None else None
                dev.mlp2_ce_backward(xd._handle, hd._handle, self.dz1_pre._handle, self.dl_pre._handle, dw1._handle, hh(db1),
                                     dw2._handle, hh(db2), self.rowloss._handle, self.loss._handle, M, K, HD, P)
                hn._pre = (self.dz1_pre, dw1, db1)
                self._grads = (grad, self.dz1_pre, dw2, db2)
            _, dz1, dw2, db2 = self._grads
            return dz1 if node is self.x else (dw2 if node is self.w else db2)
        return super().grad_fn(node, grad)


def mlp2_ce_train(x: Tensor, fc1, fc2, target: Tensor):
    """criterion(fc2(F.relu(fc1(x))), target) with CrossEntropyLoss -- the forward pass of test/MLP_MNIST_cuda.py:44-59 -- as
    ONE launch that also leaves d loss / d logits and d loss / d (fc1 output) behind (dfhip_mlp2_ce_forward).  Returns the
    three nodes the unfused calls would have put on the tape (relu(fc1(x)), the logits, the loss) or None when the device has
    no such kernel for the shapes."""
    x = realize(x)
    mod = x.device.mod
    w1, b1, w2, b2 = fc1.weight, fc1.bias, fc2.weight, fc2.bias
    if not (_fused(x) and x.ndim == 2 and hasattr(mod, "mlp2_ce_forward") and is_grad_enable() and not x.requires_grad
            and w1.requires_grad and w2.requires_grad and not target.requires_grad and target.ndim == 2
            and int(w1.shape[0]) == int(x.shape[1]) and int(w2.shape[0]) == int(w1.shape[1])
            and tuple(int(v) for v in target.shape) == (int(x.shape[0]), int(w2.shape[1])) and target.device == x.device):
        return None
    M, K = (int(v) for v in x.shape)
    HD, P = (int(v) for v in w2.shape)
    dev = x.device
    xd, w1d, w2d, td = x.data.compact(), w1.data.compact(), w2.data.compact(), target.data.compact()
    z, h, dz1 = _new((M, HD), dev), _new((M, HD), dev), _new((M, HD), dev)
    out, dl, rowloss = _new((M, P), dev), _new((M, P), dev), _new((M,), dev)
    hb = lambda t: t.data.compact()._handle if t is not None else None
    if not dev.mlp2_ce_forward(xd._handle, w1d._handle, hb(b1), w2d._handle, hb(b2), td._handle, z._handle, h._handle,
                               out._handle, dl._handle, dz1._handle, rowloss._handle, M, K, HD, P):
        return None
    loss_arr = mod.deferred_array(1)
    loss = BackendTensor.make((1,), device=dev, handle=loss_arr)
    rl = rowloss._handle
    loss_arr.owe(lambda: mod.loss_finish(rl, loss_arr, M))
    hn = _MlpHidden.__new__(_MlpHidden)
    hn.x, hn.w, hn.b = x, w1, b1
    hn._saved = (xd, w1d, z)
    hn._grads = None
    _FusedNode.__init__(hn, h, [x, w1, b1])
    ln = _MlpLogits.__new__(_MlpLogits)
    ln.x, ln.w, ln.b = hn, w2, b2
    ln._saved = (h, w2d)
    ln._grads = None
    ln.dl_pre, ln.dz1_pre, ln.rowloss, ln.loss = dl, dz1, rowloss, loss
    _FusedNode.__init__(ln, out, [hn, w2, b2])
    return hn, ln, _HeadCELoss(ln, out, td, loss, dl)


def linear_relu(input: Tensor, weight: Tensor, bias=None):
    """F.relu(F.linear(input, weight, bias)); one node on a fused device when the layer is small enough for the
    one-launch kernels."""
    input = realize(input)
    if _fused(input) and is_grad_enable():
        node = _LinearReluFused.build(input, weight, bias)
        if node is not None:
            return node
    return relu(linear(input, weight, bias))


class tanh(tensor.UnaryOperator):
    def forward(self, input):
        return backend_api.tanh(input.data)

    def grad_fn(self, input, grad):
        return (1 - self.data ** 2) * grad


def leaky_relu(input: Tensor, negative_slope: float):
    return tensor.maximum(input, input * negative_slope)


def softmax(input: Tensor, dim=None, keepdims=False):
    dim = 1 if dim is None else dim
    shifted = input - tensor.max(input, dim, True)
    e = tensor.exp(shifted)
    return e / tensor.sum(e, dim, True)


def log_softmax(input: Tensor, dim=None, keepdims=False):
    dim = 1 if dim is None else dim
    shifted = input - tensor.max(input, dim, True)
    return shifted - tensor.log(tensor.sum(tensor.exp(shifted), dim, True))


# ---- losses --------------------------------------------------------------------------------------
def nll_loss(input: Tensor, target: Tensor, reduction: str = 'mean'):
    nll = -input * target
    if reduction == 'mean':
        return tensor.mean(nll)
    if reduction == 'sum':
        return tensor.sum(nll)
    assert 0, "reduction must be mean or sum."


def mse_loss(input: Tensor, target: Tensor, reduction: str = 'mean'):
    sq = tensor.square(input - target)
    if reduction == 'mean':
        return tensor.mean(sq)
    if reduction == 'sum':
        return tensor.sum(sq)
    assert 0, "reduction must be mean or sum."


class _CrossEntropyFused(_FusedNode):
    def __init__(self, logits, target):
        self.ld = logits.data.compact()
        self.td = target.data.compact()
        rows, cols = self.ld.shape
        out = _new((1,), logits.device)
        # for a classifier-sized problem the forward kernel also leaves d loss / d logits for an upstream gradient
        # of exactly 1 behind -- what loss.backward() seeds (tensor.py:428) -- and grad_fn then launches nothing
        self.unit_grad = None
        fg = getattr(logits.device.mod, "softmax_ce_forward_grad", None)
        if fg is not None and is_grad_enable() and logits.requires_grad:
            dl = _new((rows, cols), logits.device)
            if fg(self.ld._handle, self.td._handle, out._handle, dl._handle, rows, cols):
                self.unit_grad = dl
        else:
            logits.device.softmax_ce_forward(self.ld._handle, self.td._handle, out._handle, rows, cols)
        super().__init__(out, [logits])

    def grad_fn(self, node, grad):
        if self.unit_grad is not None and getattr(grad, "is_unit_seed", False):
            return self.unit_grad
        rows, cols = self.ld.shape
        out = _new((rows, cols), grad.device)
        grad.device.softmax_ce_backward(self.ld._handle, self.td._handle, grad.compact()._handle,
                                        out._handle, rows, cols)
        return out


_lazy_head_ce = None       # nn/lazy.py: cross_entropy(fc(mean(mean(x, 2), 2)), target) as one launch, or None


def cross_entropy(input: Tensor, target: Tensor, reduction: str = 'mean', dim: int = 1):
    if _lazy_head_ce is not None and reduction == 'mean' and dim == 1:
        loss = _lazy_head_ce(input, target)
        if loss is not None:
            return loss
    input = realize(input)
    if (_fused(input) and reduction == 'mean' and dim == 1 and input.ndim == 2
            and not target.requires_grad):
        return _CrossEntropyFused(input, target)
    m = tensor.max(input, dim, True)
    shifted = input - m
    lse = tensor.log(tensor.sum(tensor.exp(shifted), dim, True))
    nll = -(shifted - lse) * target
    if reduction == 'mean':
        per_sample = tensor.sum(nll, dim, True)
        return tensor.sum(per_sample) * (1.0 / input.shape[0])
    if reduction == 'sum':
        return tensor.sum(nll)
    assert 0, "reduction must be mean or sum."


# ---- im2col building blocks (primitive path; functional.py:249-313) ---------------------------------
class _im2col2d(tensor.UnaryOperator):
    def __init__(self, x, kernel_size, stride):
        self.N, self.C, self.H, self.W = x.shape
        self.k, self.s = kernel_size, stride
        self.oh = (self.H - self.k) // self.s + 1
        self.ow = (self.W - self.k) // self.s + 1
        super().__init__(x)

    def forward(self, x):
        xp = x.data
        col = backend_api.zeros((self.N, self.C, self.k, self.k, self.oh, self.ow), device=x.device)
        for i in range(self.k):
            i_hi = i + self.oh * self.s
            for j in range(self.k):
                j_hi = j + self.ow * self.s
                col[:, :, i, j, :, :] = xp[:, :, i:i_hi:self.s, j:j_hi:self.s]
        return col

    def grad_fn(self, x, grad):
        gx = backend_api.zeros((self.N, self.C, self.H, self.W), device=x.device)
        for i in range(self.k):
            i_hi = i + self.oh * self.s
            for j in range(self.k):
                j_hi = j + self.ow * self.s
                # assignment, not accumulation: overlapping windows overwrite (functional.py:292-293)
                gx[:, :, i:i_hi:self.s, j:j_hi:self.s] = grad[:, :, i, j, :, :]
        return gx


class _pad2d(tensor.UnaryOperator):
    def __init__(self, x, pad_width=0):
        self.p = pad_width
        super().__init__(x)

    def forward(self, x):
        return backend_api.pad(x.data, ((0, 0), (0, 0), (self.p, self.p), (self.p, self.p)))

    def grad_fn(self, x, grad):
        if self.p == 0:
            return grad
        return grad[:, :, self.p:-self.p, self.p:-self.p]


# ---- conv2d ---------------------------------------------------------------------------------------
class _Conv2dFused(_FusedNode):
    def __init__(self, x, w, bias, padding, stride):
        self.x, self.w, self.bias = x, w, bias
        N, C, H, W = (int(v) for v in x.shape)
        O, _, k, _ = (int(v) for v in w.shape)
        oh = (H + 2 * padding - k) // stride + 1
        ow = (W + 2 * padding - k) // stride + 1
        self.geo = (N, C, H, W, O, k, int(stride), int(padding))
        self.xd, self.wd = x.data.compact(), w.data.compact()
        out = _new((N, O, oh, ow), x.device)
        bh = bias.data.compact()._handle if bias is not None else None
        # BatchNorm partials of the output, left behind by the convolution's epilogue: a
        # BatchNorm2d that consumes this node skips its statistics pass (bn_partials = None when
        # the shape was served by a kernel without that epilogue)
        self.bn_partials = None
        stats_fn = getattr(x.device.mod, "conv2d_forward_stats", None)
        if stats_fn is not None and is_grad_enable():
            part = _new((_BN_PARTIAL_ROWS, O, 3), x.device)
            rows = stats_fn(self.xd._handle, self.wd._handle, bh, out._handle, *self.geo,
                            part._handle, _BN_PARTIAL_ROWS)
            if rows > 0:
                self.bn_partials = (part, rows)
        else:
            x.device.conv2d_forward(self.xd._handle, self.wd._handle, bh, out._handle, *self.geo)
        super().__init__(out, [x, w, bias])

    @classmethod
    def pair_s2(cls, x, conv1, conv_d):
        """(conv1(x), conv_d(x)) for a 3x3 / stride 2 / pad 1 `conv1` and a 1x1 / stride 2 / pad 0 `conv_d` without
        biases that read the same tensor -- a residual block and its shortcut (test/ResNet_CIFAR10_cuda.py:31-47) -- from
        ONE launch (the shortcut rides as a tenth tap of the 3x3 loop); two ordinary nodes come back, each with its own
        grad_fn.  None when that form does not apply: the caller then calls the two modules."""
        x = realize(x)
        dev = x.device
        fn = getattr(dev.mod, "conv2d_forward_s2_pair", None)
        if (fn is None or len(x.shape) != 4 or conv1.bias is not None or conv_d.bias is not None
                or (int(conv1.kernel_size), int(conv1.stride), int(conv1.padding)) != (3, 2, 1)
                or (int(conv_d.kernel_size), int(conv_d.stride), int(conv_d.padding)) != (1, 2, 0)
                or tuple(conv1.weight.shape[:2]) != tuple(conv_d.weight.shape[:2])):
            return None
        N, C, H, W = (int(v) for v in x.shape)
        O = int(conv1.weight.shape[0])
        if H % 2 or W % 2:
            return None
        xd = x.data.compact()
        w1, wd = conv1.weight.data.compact(), conv_d.weight.data.compact()
        y1, yd = _new((N, O, H // 2, W // 2), dev), _new((N, O, H // 2, W // 2), dev)
        stats = is_grad_enable()
        p1 = _new((_BN_PARTIAL_ROWS, O, 3), dev) if stats else None
        pd = _new((_BN_PARTIAL_ROWS, O, 3), dev) if stats else None
        h = lambda t: t._handle if t is not None else None
        done, rows = fn(xd._handle, w1._handle, wd._handle, y1._handle, yd._handle, N, C, H, W, O, h(p1), h(pd),
                        _BN_PARTIAL_ROWS)
        if not done:
            return None
        nodes = []
        # (the shortcut's node first: the backward sweep then reaches conv1's node first, which forms both input
        # gradients in one launch -- shortcut_sibling)
        for out, wt, wdat, part, geo in ((yd, conv_d.weight, wd, pd, (N, C, H, W, O, 1, 2, 0)),
                                         (y1, conv1.weight, w1, p1, (N, C, H, W, O, 3, 2, 1))):
            node = cls.__new__(cls)
            node.x, node.w, node.bias = x, wt, None
            node.geo, node.xd, node.wd = geo, xd, wdat
            node.bn_partials = (part, rows) if (part is not None and rows > 0) else None
            _FusedNode.__init__(node, out, [x, wt, None])
            nodes.append(node)
        nodes[0]._sibling_host = weakref.ref(nodes[1])     # (the shortcut's node: its input gradient rides on conv1's launch)
        return nodes[1], nodes[0]

    # the engine may hand over the gradient the input already holds (tensor.py backward): the sum
    # is then formed in the epilogue of the input-gradient kernel
    fuses_accumulation = True
    # residual block with a convolution shortcut: conv1's node points at the shortcut convolution's node (both read
    # the block input).  The backward sweep reaches conv1 first; if the shortcut's output gradient is complete by
    # then, ONE launch forms both input gradients and their sum, and the shortcut's node hands nothing back for x.
    shortcut_sibling = None
    _dx_delivered = False
    # the stem's pooling node computed this node's weight gradient inside its own backward launch
    # (_BnReluPoolFused.grad_fn -> dfhip_stem_backward): grad_fn hands it out, the gradient it is called with is a
    # placeholder
    _dw_ready = None
    _dw_from_stem = False        # (diagnostic: the last weight gradient handed out came from the stem's launch)

    def _pending_ok(self):
        """this node's input- and weight-gradient kernels can take a pending BatchNorm backward as their dY"""
        mod = self.device.mod
        if self.bias is not None:
            return False
        host = self._sibling_host() if getattr(self, "_sibling_host", None) is not None else None
        if host is not None or self.shortcut_sibling is not None:
            return False             # (the pair launch of a block with a convolution shortcut takes no pending operand)
        dg = mod.conv2d_bn_bwd_supported(*self.geo)[0] if self.x.requires_grad else True
        wg = mod.conv2d_bn_bwd_supported(*self.geo)[1] if self.w.requires_grad else True
        return dg and wg

    def grad_fn(self, node, grad, accumulated=None):
        pend = getattr(grad, "bn_pending", None)
        g = grad.compact()
        dev = g.device
        h = lambda t: t._handle if t is not None else None
        if node is self.x:
            if self._dx_delivered:
                return None                       # conv1's launch already carried this node's input gradient
            sib = self.shortcut_sibling
            N, C, H, W, O = self.geo[:5]
            add = accumulated.compact() if accumulated is not None else None
            if (sib is not None and sib.x is self.x and sib.grad is not None and not sib._dx_delivered
                    and hasattr(dev.mod, "conv2d_dgrad_pair") and self.geo[5:] == (3, 2, 1) and sib.geo[5:] == (1, 2, 0)
                    and sib.geo[:5] == self.geo[:5]):
                sg = sib.grad.data if isinstance(sib.grad, Tensor) else sib.grad
                spend = getattr(sg, "bn_pending", None)
                out = _new(self.xd.shape, dev)
                targets = _stats_targets(self.x, self, sib)
                ok, _, cap = dev.mod.conv2d_bn_bwd_supported(*self.geo, pair=True)
                if (pend is not None or spend is not None or targets) and ok:
                    rows = _new_rows(cap, C, 1 + len(targets), dev) if targets else None
                    nrows = dev.conv2d_dgrad_pair_bn(g._handle, self.wd._handle, sg.compact()._handle, sib.wd._handle, h(add),
                                                     out._handle, N, C, H, W, O, pend, spend, targets, h(rows), cap)
                    if pend is not None:
                        pend.table_ready = True
                    if spend is not None:
                        spend.table_ready = True
                    if targets:
                        _tag_rows(out, self.x, rows, nrows, cap, 1 + len(targets))
                    sib._dx_delivered = True
                    return out
                if pend is not None:
                    g = _pending_materialise(pend, g)
                    self._plain_grad = (grad, g)
                    pend = None                   # (dz exists now: the single-launch path below must not apply it again)
                if spend is not None:
                    sg = _pending_materialise(spend, sg)
                    sib.grad = sg                 # (the shortcut's weight gradient then reads the plain tensor)
                    spend = None
                if dev.conv2d_dgrad_pair(g._handle, self.wd._handle, sg.compact()._handle, sib.wd._handle, h(add),
                                         out._handle, N, C, H, W, O):
                    sib._dx_delivered = True
                    return out
            out = _new(self.xd.shape, dev)
            targets = _stats_targets(self.x, self)
            ok, _, cap = dev.mod.conv2d_bn_bwd_supported(*self.geo)
            if (pend is not None or targets) and ok:
                rows = _new_rows(cap, C, 1 + len(targets), dev) if targets else None
                nrows = dev.conv2d_dgrad_bn(g._handle, self.wd._handle, h(add), out._handle, *self.geo, pend, targets,
                                            h(rows), cap)
                if pend is not None:
                    pend.table_ready = True
                if targets:
                    _tag_rows(out, self.x, rows, nrows, cap, 1 + len(targets))
                return out
            if pend is not None:
                g = self._plain_of(grad, pend)
            if add is not None:
                dev.conv2d_dgrad_add(g._handle, self.wd._handle, add._handle, out._handle, *self.geo)
            else:
                dev.conv2d_dgrad(g._handle, self.wd._handle, out._handle, *self.geo)
            return out
        if node is self.w:
            if self._dw_ready is not None:
                dw, placeholder = self._dw_ready
                self._dw_ready = None             # consumed: a second backward() over a retained graph starts afresh
                if grad is placeholder:
                    self._dw_from_stem = True
                    return dw
                # cannot happen while the pooling node checks that it is the only consumer (the placeholder holds
                # no values to fall back on): refuse rather than return a gradient without the other contribution
                raise RuntimeError("stem backward: the convolution's output gradient is not the placeholder its "
                                   "pooling node handed out (another consumer?)")
            out = _grad_out(self.w, self.wd.shape, dev)
            if pend is not None:
                _pending_table(pend, dev)         # (written by the input-gradient launch where there was one)
                if dev.conv2d_wgrad_bn(self.xd._handle, None, g._handle, pend.z, pend.table, out._handle, *self.geo):
                    return out
                g = self._plain_of(grad, pend)
            dev.conv2d_wgrad(self.xd._handle, g._handle, out._handle, *self.geo)
            return out
        if pend is not None:
            g = self._plain_of(grad, pend)
        out = _grad_out(self.bias, self.bias.shape, dev)
        dev.reduce_to_shape(g._handle, out._handle, g.shape, self.bias.shape)
        return out

    _plain_grad = None

    def _plain_of(self, grad, pend):
        """the pending gradient written out, once per gradient tensor"""
        if self._plain_grad is None or self._plain_grad[0] is not grad:
            self._plain_grad = (grad, _pending_materialise(pend, grad))
        return self._plain_grad[1]


def conv2d(x: Tensor, kernel: Tensor, padding: int = 0, stride: int = 1, bias=None):
    """x (N,C,H,W), kernel (O,C,k,k) -> (N,O,oh,ow).  `bias` (1,O,1,1) is an extension used by
    nn.Conv2d so the fused kernel adds it in its epilogue; the reference adds it in
    Conv2d.forward (conv.py:104-108)."""
    x = realize(x)
    if _fused(x):
        return _Conv2dFused(x, kernel, bias, padding, stride)
    N = x.shape[0]
    O, _, k, _ = kernel.shape
    col = _im2col2d(_pad2d(x, padding), k, stride)
    oh, ow = col.shape[-2:]
    col = col.transpose(0, 4, 5, 1, 2, 3).reshape(N * oh * ow, -1)
    out = col @ kernel.reshape(O, -1).T
    out = out.reshape(N, oh, ow, -1).transpose(0, 3, 1, 2)
    return out if bias is None else out + bias


class _MeanHWFused(_FusedNode):
    """tensor.mean(tensor.mean(x, axis=2), axis=2) of an (N, C, H, W) tensor as one reduction and one
    broadcast: the two means of the reference divide by the TOTAL element count of their input
    (SURVEY F5), so the value is sum_hw(x) / (N*C*H*W) / (N*C*W) and the gradient is g / (H*W)."""

    def __init__(self, x):
        self.x = x
        N, C, H, W = (int(v) for v in x.shape)
        self.geo = (N, C, H, W)
        out = _new((N, C), x.device)
        scale = 1.0 / (float(N * C * H * W) * float(N * C * W))
        x.device.reduce_axis(x.data.compact()._handle, out._handle, N * C, H * W, 1, "sum", scale)
        super().__init__(out, [x])

    def grad_fn(self, node, grad):
        N, C, H, W = self.geo
        out = _new((N, C, H, W), grad.device)
        # mean.grad_fn scales by out.size / in.size at each of the two steps: (N*C*W)/(N*C*H*W) * (N*C)/(N*C*W)
        grad.device.broadcast_axis(grad.compact()._handle, out._handle, N * C, H * W, 1, 1.0 / float(H * W))
        return out


class _PoolLinearFused(_FusedNode):
    """fc(mean(mean(x, axis=2), axis=2)) -- the classifier head of test/ResNet_CIFAR10_cuda.py:96-99 -- as one node:
    one launch forward (pooled features kept for the backward pass), one launch backward (dx broadcast over the
    pixels, dW, db).  Values as _MeanHWFused followed by _LinearFused: the means divide by the TOTAL element count
    of their input and their gradients scale by out.size / in.size."""

    def __init__(self, x, w, b):
        self.x, self.w, self.b = x, w, b
        N, C, H, W = (int(v) for v in x.shape)
        P = int(w.shape[1])
        self.geo = (N, C, H * W, P)
        dev = x.device
        self.wd = w.data.compact()
        self.pooled = _new((N, C), dev)
        out = _new((N, P), dev)
        scale = 1.0 / (float(N * C * H * W) * float(N * C * W))
        dev.pool_linear_forward(x.data.compact()._handle, self.wd._handle,
                                b.data.compact()._handle if b is not None else None, self.pooled._handle,
                                out._handle, N, C, H * W, P, scale)
        self._grads = None
        super().__init__(out, [x, w, b])

    def grad_fn(self, node, grad):
        if self._grads is None or self._grads[0] is not grad:
            N, C, HW, P = self.geo
            dev = grad.device
            dx = _new(self.x.shape, dev) if self.x.requires_grad else None
            dw = _grad_out(self.w, self.wd.shape, dev) if self.w.requires_grad else None
            db = _grad_out(self.b, self.b.shape, dev) if (self.b is not None and self.b.requires_grad) else None
            h = lambda t: t._handle if t is not None else None
            dev.pool_linear_backward(grad.compact()._handle, self.pooled._handle, self.wd._handle, h(dx), h(dw), h(db),
                                     N, C, HW, P, 1.0 / float(HW))
            self._grads = (grad, dx, dw, db)
        _, dx, dw, db = self._grads
        return dx if node is self.x else (dw if node is self.w else db)


class _HeadLogits(_FusedNode):
    """The logits of `head_ce_train`: the node _PoolLinearFused would be, its arrays filled by the head + loss launch.  When the
    gradient that arrives is the d loss / d logits that launch left behind (nothing else used the logits), the input gradient
    is the one it left behind too and only the sums over the images (dW, db) remain -- extra workgroups of the sweep's grouped
    slab sum on the hip device (dfhip_head_grads), a launch otherwise; any other gradient takes _PoolLinearFused's launch."""

    def __init__(self, x, w, b, wd, pooled, out, dl, dx, rowloss, loss):
        self.x, self.w, self.b, self.wd, self.pooled = x, w, b, wd, pooled
        self.rowloss, self.loss = rowloss, loss
        N, C, H, W = (int(v) for v in x.shape)
        self.geo = (N, C, H * W, int(w.shape[1]))
        self.dl_pre, self.dx_pre = dl, dx
        self._grads = None
        super().__init__(out, [x, w, b])

    def grad_fn(self, node, grad):
        if self._grads is None or self._grads[0] is not grad:
            N, C, HW, P = self.geo
            dev = grad.device
            dw = _grad_out(self.w, self.wd.shape, dev) if self.w.requires_grad else None
            db = _grad_out(self.b, self.b.shape, dev) if (self.b is not None and self.b.requires_grad) else None
            h = lambda t: t._handle if t is not None else None
            if grad is self.dl_pre and (self.dx_pre is not None or not self.x.requires_grad):
                dx = self.dx_pre
                # (dW, db and the mean of the row losses: extra workgroups of the sweep's slab sum inside a group)
                dev.head_grads(self.dl_pre._handle, self.pooled._handle, h(dw), h(db), N, C, P,
                               self.rowloss._handle, self.loss._handle)
            else:
                dx = _new(self.x.shape, dev) if self.x.requires_grad else None
                dev.pool_linear_backward(grad.compact()._handle, self.pooled._handle, self.wd._handle, h(dx), h(dw), h(db),
                                         N, C, HW, P, 1.0 / float(HW))
            self._grads = (grad, dx, dw, db)
        _, dx, dw, db = self._grads
        return dx if node is self.x else (dw if node is self.w else db)


class _HeadCELoss(_FusedNode):
    """The loss of `head_ce_train`: the node _CrossEntropyFused would be, with d loss / d logits for the unit seed on file."""

    def __init__(self, logits, ld, td, out, dl):
        self.ld, self.td, self.unit_grad = ld, td, dl
        super().__init__(out, [logits])

    grad_fn = _CrossEntropyFused.grad_fn


def head_ce_train(x: Tensor, fc, target: Tensor):
    """criterion(fc(mean(mean(x, 2), 2)), target) with CrossEntropyLoss -- the end of the forward pass of
    test/ResNet_CIFAR10_cuda.py (:96-99, :186) -- as ONE launch that also leaves d loss / d logits and the head's input
    gradient behind (everything of the backward pass that belongs to one image; dfhip_head_ce_forward).  Returns
    (logits node, loss node) -- the two nodes the unfused calls would have put on the tape, so that whatever else reads
    the logits differentiates correctly -- or None when the device has no such kernel for the shape."""
    x = realize(x)
    mod = x.device.mod
    if not (_fused(x) and len(x.shape) == 4 and hasattr(mod, "head_ce_forward") and is_grad_enable() and x.requires_grad
            and not target.requires_grad and target.ndim == 2 and int(fc.weight.shape[1]) <= 16
            and int(fc.weight.shape[0]) == int(x.shape[1]) and int(target.shape[0]) == int(x.shape[0])
            and int(target.shape[1]) == int(fc.weight.shape[1]) and target.device == x.device):
        return None
    N, C, H, W = (int(v) for v in x.shape)
    P = int(fc.weight.shape[1])
    dev = x.device
    w, b = fc.weight, fc.bias
    wd = w.data.compact()
    td = target.data.compact()
    pooled, out, dl = _new((N, C), dev), _new((N, P), dev), _new((N, P), dev)
    rowloss, dx = _new((N,), dev), _new(x.shape, dev)
    # the loss is the mean of the row losses the launch leaves behind: added up by the backward pass (head_grads), or by a
    # launch of its own if somebody reads it first -- the array says so (hip_device.DeferredHipArray)
    loss_arr = mod.deferred_array(1)
    loss = BackendTensor.make((1,), device=dev, handle=loss_arr)
    scale = 1.0 / (float(N * C * H * W) * float(N * C * W))
    if not dev.head_ce_forward(x.data.compact()._handle, wd._handle, b.data.compact()._handle if b is not None else None,
                               td._handle, pooled._handle, out._handle, loss._handle, dl._handle, dx._handle, rowloss._handle,
                               N, C, H * W, P, scale, 1.0 / float(H * W)):
        return None
    rl = rowloss._handle
    loss_arr.owe(lambda: mod.loss_finish(rl, loss_arr, N))
    logits = _HeadLogits(x, w, b, wd, pooled, out, dl, dx, rowloss, loss)
    return logits, _HeadCELoss(logits, out, td, loss, dl)


class _DropLinLogits(_FusedNode):
    """The logits of `droplin_ce_train` -- fc(drop(x)) -- with its arrays filled by the Dropout + Linear + loss launch.  When
    the gradient that arrives is the d loss / d logits that launch left behind, the gradient at Dropout's input is the one it
    left behind too and only dW, db remain (dfhip_head_grads: extra workgroups of the sweep's grouped slab sum inside a group);
    any other gradient takes the Linear's backward launch and Dropout's."""

    def __init__(self, x, mask, keep, w, b, wd, xd, out, dl, dx, rowloss, loss):
        self.x, self.mask, self.keep, self.w, self.b, self.wd, self.xd = x, mask, float(keep), w, b, wd, xd
        self.rowloss, self.loss = rowloss, loss
        self.geo = (int(x.shape[0]), int(x.shape[1]), int(w.shape[1]))
        self.dl_pre, self.dx_pre = dl, dx
        self._grads = None
        super().__init__(out, [x, w, b])

    def grad_fn(self, node, grad):
        if self._grads is None or self._grads[0] is not grad:
            N, C, P = self.geo
            dev = grad.device
            dw = _grad_out(self.w, self.wd.shape, dev) if self.w.requires_grad else None
            db = _grad_out(self.b, self.b.shape, dev) if (self.b is not None and self.b.requires_grad) else None
            h = lambda t: t._handle if t is not None else None
            if grad is self.dl_pre and (self.dx_pre is not None or not self.x.requires_grad):
                dx = self.dx_pre
                dev.head_grads(self.dl_pre._handle, self.xd._handle, h(dw), h(db), N, C, P, self.rowloss._handle,
                               self.loss._handle)
            else:
                dxd = _new(self.x.shape, dev) if self.x.requires_grad else None
                dev.pool_linear_backward(grad.compact()._handle, self.xd._handle, self.wd._handle, h(dxd), h(dw), h(db),
                                         N, C, 1, P, 1.0)
                dx = None
                if dxd is not None:
                    dx = _new(self.x.shape, dev)
                    dev.mod.dropout_apply(dxd._handle, self.mask._handle, dx._handle, dxd._handle.size, self.keep, True)
            self._grads = (grad, dx, dw, db)
        _, dx, dw, db = self._grads
        return dx if node is self.x else (dw if node is self.w else db)


def droplin_ce_train(x: Tensor, mask, keep, fc, target: Tensor):
    """criterion(fc(drop(x)), target) with CrossEntropyLoss for a two-dimensional x -- the end of the forward pass of
    test/CNN_CIFAR10_cuda.py (:104-107 and its criterion) -- as ONE launch that also leaves d loss / d logits and the gradient
    at Dropout's input behind (dfhip_droplin_ce_forward).  Returns (logits node, loss node) or None when the device has no such
    kernel for the shape."""
    x = realize(x)
    mod = x.device.mod
    if not (_fused(x) and len(x.shape) == 2 and hasattr(mod, "droplin_ce_forward") and is_grad_enable() and x.requires_grad
            and not target.requires_grad and target.ndim == 2 and int(fc.weight.shape[1]) <= 16
            and int(fc.weight.shape[0]) == int(x.shape[1]) and int(x.shape[1]) <= 2048
            and int(target.shape[0]) == int(x.shape[0]) and int(target.shape[1]) == int(fc.weight.shape[1])
            and target.device == x.device and tuple(int(v) for v in mask.shape) == tuple(int(v) for v in x.shape)):
        return None
    N, C = (int(v) for v in x.shape)
    P = int(fc.weight.shape[1])
    dev = x.device
    w, b = fc.weight, fc.bias
    wd = w.data.compact()
    td = target.data.compact()
    md = mask.compact()
    xd, out, dl = _new((N, C), dev), _new((N, P), dev), _new((N, P), dev)
    rowloss, dx = _new((N,), dev), _new(x.shape, dev)
    loss_arr = mod.deferred_array(1)                    # (the mean of the row losses: see head_ce_train)
    loss = BackendTensor.make((1,), device=dev, handle=loss_arr)
    if not mod.droplin_ce_forward(x.data.compact()._handle, md._handle, float(keep), wd._handle,
                                  b.data.compact()._handle if b is not None else None, td._handle, xd._handle, out._handle,
                                  loss._handle, dl._handle, dx._handle, rowloss._handle, N, C, P):
        return None
    rl = rowloss._handle
    loss_arr.owe(lambda: mod.loss_finish(rl, loss_arr, N))
    logits = _DropLinLogits(x, md, keep, w, b, wd, xd, out, dl, dx, rowloss, loss)
    return logits, _HeadCELoss(logits, out, td, loss, dl)


def mean_hw_linear(x: Tensor, fc) -> Tensor:
    """fc(mean_hw(x)) for an nn.Linear `fc`; one node on a fused device when the layer is classifier-sized."""
    x = realize(x)
    mod = x.device.mod
    if (_fused(x) and len(x.shape) == 4 and hasattr(mod, "pool_linear_forward") and int(fc.weight.shape[1]) <= 16
            and int(fc.weight.shape[0]) == int(x.shape[1])):
        return _PoolLinearFused(x, fc.weight, fc.bias)
    return fc(mean_hw(x))


def mean_hw(x: Tensor) -> Tensor:
    """mean over H then over W with the reference's `mean(axis)` semantics (two calls of
    tensor.mean(x, axis=2)); one kernel each way on a fused device."""
    x = realize(x)
    if _fused(x) and len(x.shape) == 4 and hasattr(x.device.mod, "reduce_axis"):
        return _MeanHWFused(x)
    return tensor.mean(tensor.mean(x, axis=2), axis=2)


# ---- pooling ------------------------------------------------------------------------------------------
class _MaxPool2dFused(_FusedNode):
    def __init__(self, x, k):
        N, C, H, W = (int(v) for v in x.shape)
        self.geo = (N * C, H, W, int(k))
        self.xd = x.data.compact()
        out = _new((N, C, (H - k) // k + 1, (W - k) // k + 1), x.device)
        x.device.maxpool2d_forward(self.xd._handle, out._handle, *self.geo)
        super().__init__(out, [x])

    def grad_fn(self, node, grad):
        out = _new(self.xd.shape, grad.device)
        grad.device.maxpool2d_backward(self.xd._handle, self.data._handle, grad.compact()._handle,
                                       out._handle, *self.geo)
        return out


def max_pool2d(x: Tensor, kernel_size: int, stride: int, padding=0):
    x = realize(x)
    if _fused(x) and padding == 0 and stride == kernel_size:
        return _MaxPool2dFused(x, kernel_size)
    N, C = x.shape[0], x.shape[1]
    col = _im2col2d(_pad2d(x, padding), kernel_size, stride)
    oh, ow = col.shape[-2:]
    col = col.transpose(0, 4, 5, 1, 2, 3).reshape(-1, kernel_size * kernel_size)
    out = col.max(1)
    return out.reshape(N, oh, ow, C).transpose(0, 3, 1, 2)


def avg_pool2d(x: Tensor, kernel_size: int, stride: int, padding=0):
    x = realize(x)
    N, C = x.shape[0], x.shape[1]
    col = _im2col2d(_pad2d(x, padding), kernel_size, stride)
    oh, ow = col.shape[-2:]
    col = col.transpose(0, 4, 5, 1, 2, 3).reshape(-1, kernel_size * kernel_size)
    # mean over the window = sum / k^2 (padding zeros included, functional.py:395-396)
    out = tensor.sum(col, 1, True) * (1.0 / (kernel_size * kernel_size))
    return out.reshape(N, oh, ow, C).transpose(0, 3, 1, 2)


# ---- batch norm (used by nn.BatchNorm2d; the reference inlines it in batchnorm.py:30-55) ----------------
class _BatchNorm2dFused(_FusedNode):
    """Training-mode BatchNorm2d (+ optional residual add: y = bn(x) + residual, the tail of a
    residual block).  When x is a convolution output that carries BatchNorm partials the
    statistics come from those and x is read once."""

    def __init__(self, x, weight, bias, running_mean, running_var, eps, momentum, residual=None):
        self.x, self.weight, self.bias, self.residual = x, weight, bias, residual
        N, C, H, W = (int(v) for v in x.shape)
        self.geo = (N, C, H * W)
        self.eps = float(eps)
        dev = x.device
        self.xd = x.data.compact()
        self.wd = weight.data.compact() if weight is not None else None
        out = _new(x.shape, dev)
        self.mean, self.var = _new((1, C, 1, 1), dev), _new((1, C, 1, 1), dev)
        h = lambda t: t._handle if t is not None else None
        bh = h(bias.data.compact()) if bias is not None else None
        rm = h(running_mean.data) if running_mean is not None else None
        rv = h(running_var.data) if running_var is not None else None
        partials = getattr(x, "bn_partials", None)
        if partials is not None and hasattr(dev.mod, "bn2d_forward_partials"):
            part, rows = partials
            rd = residual.data.compact() if residual is not None else None
            dev.bn2d_forward_partials(self.xd._handle, part._handle, rows, h(self.wd), bh, h(rd), out._handle,
                                      self.mean._handle, self.var._handle, rm, rv, N, C, H * W, self.eps,
                                      float(momentum))
        else:
            dev.bn2d_forward_train(self.xd._handle, h(self.wd), bh, out._handle, self.mean._handle,
                                   self.var._handle, rm, rv, N, C, H * W, self.eps, float(momentum))
            if residual is not None:
                out = out + residual.data
        self._grads = None
        super().__init__(out, [x, weight, bias, residual])

    _defer = None

    def _deferred_targets(self):
        """[(z, mean)] when this layer's backward pass can ride on its neighbours' kernels (decided once), else None"""
        if self._defer is None:
            self._defer = bool(self.weight is not None and self.x.requires_grad and _accepts_pending(self.x))
        return [(self.xd._handle, self.mean._handle)] if self._defer else None

    def grad_fn(self, node, grad):
        if self._grads is None or self._grads[0] is not grad:
            dev = grad.device
            N, C, HW = self.geo
            need_dx = self.x.requires_grad
            dg, db = _grad_out(self.weight, (1, C, 1, 1), dev), _grad_out(self.bias, (1, C, 1, 1), dev)
            h = lambda t: t._handle if t is not None else None
            left = getattr(grad, "bn_rows", None)
            if left is not None and left.owner is self and self._defer:
                # the producer of `grad` left this layer's sums behind: the convolution in front forms dx on its loads
                table = _new((C, 4), dev)
                pend = _PendingBN(self.xd._handle, left.rows._handle, left.nrows, left.cap_rows, left.K, 1, self.mean._handle,
                                  self.var._handle, h(self.wd), self.eps, (N, C, HW), dg._handle, db._handle,
                                  table._handle, (self.xd, left.rows, self.mean, self.var, self.wd, dg, db, table, grad))
                dx = _alias(grad)
                dx.bn_pending = pend
            else:
                dx = _new(self.xd.shape, dev) if need_dx else None
                dev.bn2d_backward(grad.compact()._handle, self.xd._handle, h(self.wd), self.mean._handle,
                                  self.var._handle, h(dx), dg._handle, db._handle, N, C, HW, self.eps)
            self._grads = (grad, dx, dg, db)
        _, dx, dg, db = self._grads
        if node is self.x:
            return dx
        if node is self.residual:
            return grad
        return dg if node is self.weight else db


def batch_norm2d_train(x, weight, bias, running_mean, running_var, eps, momentum, residual=None):
    x, residual = realize(x), realize(residual)
    return _BatchNorm2dFused(x, weight, bias, running_mean, running_var, eps, momentum, residual)


class _BatchNorm2dPairFused(_FusedNode):
    """y = bn_a(xa) + bn_b(xb), both in training mode on convolution outputs that carry BatchNorm partials:
    the tail of a residual block whose shortcut is convolution + BatchNorm (batchnorm.py:30-55 twice and
    the block's "out + identity").  One forward launch (bn_b's output is never materialised) and one
    backward launch for both layers (they receive the same gradient)."""

    def __init__(self, xa, bn_a, xb, bn_b):
        self.xs, self.bns = (xa, xb), (bn_a, bn_b)
        N, C, H, W = (int(v) for v in xa.shape)
        self.geo = (N, C, H * W)
        dev = xa.device
        self.xd = [x.data.compact() for x in self.xs]
        self.wd = [bn.weight.data.compact() if bn.weight is not None else None for bn in self.bns]
        self.stats = [(_new((1, C, 1, 1), dev), _new((1, C, 1, 1), dev)) for _ in range(2)]
        out = _new(xa.shape, dev)
        h = lambda t: t._handle if t is not None else None
        sets = []
        for x, xd, wd, bn, (mean, var) in zip(self.xs, self.xd, self.wd, self.bns, self.stats):
            part, rows = x.bn_partials
            sets.append((xd._handle, part._handle, rows, h(wd),
                         h(bn.bias.data.compact()) if bn.bias is not None else None, mean._handle, var._handle,
                         h(bn.running_mean.data) if bn.running_mean is not None else None,
                         h(bn.running_var.data) if bn.running_var is not None else None, bn.eps, bn.momentum))
        dev.bn2d_forward_partials_pair(sets[0], sets[1], out._handle, N, C, H * W)
        self._grads = None
        super().__init__(out, [xa, bn_a.weight, bn_a.bias, xb, bn_b.weight, bn_b.bias])

    _defer = None

    def _deferred_targets(self):
        """[(z_a, mean_a), (z_b, mean_b)] when both layers' backward passes can ride on their neighbours' kernels"""
        if self._defer is None:
            self._defer = bool(all(bn.weight is not None for bn in self.bns) and all(x.requires_grad for x in self.xs)
                               and all(_accepts_pending(x) for x in self.xs))
        if not self._defer:
            return None
        return [(xd._handle, mean._handle) for xd, (mean, _) in zip(self.xd, self.stats)]

    def grad_fn(self, node, grad):
        if self._grads is None or self._grads[0] is not grad:
            dev = grad.device
            N, C, HW = self.geo
            h = lambda t: t._handle if t is not None else None
            res, sets = [], []
            left = getattr(grad, "bn_rows", None)
            if left is not None and left.owner is self and self._defer and left.K == 3:
                for i, (xd, wd, bn, (mean, var)) in enumerate(zip(self.xd, self.wd, self.bns, self.stats)):
                    dg, db, table = _grad_out(bn.weight, (1, C, 1, 1), dev), _grad_out(bn.bias, (1, C, 1, 1), dev), _new((C, 4), dev)
                    pend = _PendingBN(xd._handle, left.rows._handle, left.nrows, left.cap_rows, left.K, 1 + i, mean._handle, var._handle,
                                      h(wd), float(bn.eps), (N, C, HW), dg._handle, db._handle, table._handle,
                                      (xd, left.rows, mean, var, wd, dg, db, table, grad))
                    dx = _alias(grad)
                    dx.bn_pending = pend
                    res.append((dx, dg, db))
            else:
                for x, xd, wd, bn, (mean, var) in zip(self.xs, self.xd, self.wd, self.bns, self.stats):
                    dx = _new(xd.shape, dev) if x.requires_grad else None
                    dg, db = _grad_out(bn.weight, (1, C, 1, 1), dev), _grad_out(bn.bias, (1, C, 1, 1), dev)
                    res.append((dx, dg, db))
                    sets.append((xd._handle, h(wd), mean._handle, var._handle, h(dx), dg._handle, db._handle, bn.eps))
                dev.bn2d_backward_pair(grad.compact()._handle, sets[0], sets[1], N, C, HW)
            self._grads = (grad, res)
        _, res = self._grads
        for i in range(2):
            if node is self.xs[i]:
                return res[i][0]
            if node is self.bns[i].weight:
                return res[i][1]
            if node is self.bns[i].bias:
                return res[i][2]
        raise RuntimeError("not a parent of this node")


class _BnConv2dFused(_FusedNode):
    """conv2d(batch_norm2d_train(x)) with NO activation in between -- `out = bn1(conv1(x)); out = conv2(out)` of
    the reference's residual block (test/ResNet_CIFAR10_cuda.py:31-35) -- as one node.  x is conv1's raw output and
    carries its BatchNorm partials; the convolution kernel combines them and normalises its operand on load, so the
    normalised tensor is never written.  Backward = the reference's chain for the same graph: weight gradient of
    the convolution (its A operand rebuilt from x and the saved scale / shift), input gradient of the convolution,
    then BatchNorm backward from x and the saved batch statistics."""

    @classmethod
    def build(cls, x, bn, conv):
        dev = x.device
        part_in = getattr(x, "bn_partials", None)
        if part_in is None or len(x.shape) != 4:
            return None
        N, C, H, W = (int(v) for v in x.shape)
        w, cbias = conv.weight, conv.bias
        O, _, k, _ = (int(v) for v in w.shape)
        stride, padding = int(conv.stride), int(conv.padding)
        oh = (H + 2 * padding - k) // stride + 1
        ow = (W + 2 * padding - k) // stride + 1
        geo = (N, C, H, W, O, k, stride, padding)
        xd, wd = x.data.compact(), w.data.compact()
        gd = bn.weight.data.compact() if bn.weight is not None else None
        bd = bn.bias.data.compact() if bn.bias is not None else None
        out = _new((N, O, oh, ow), dev)
        mean, var = _new((1, C, 1, 1), dev), _new((1, C, 1, 1), dev)
        affine = _new((C, 2), dev)
        part_out = _new((_BN_PARTIAL_ROWS, O, 3), dev)
        h = lambda t: t._handle if t is not None else None
        rm = h(bn.running_mean.data) if bn.running_mean is not None else None
        rv = h(bn.running_var.data) if bn.running_var is not None else None
        done, rows = dev.conv2d_forward_bn_input(
            xd._handle, part_in[0]._handle, part_in[1], h(gd), h(bd), float(bn.eps), float(bn.momentum),
            mean._handle, var._handle, rm, rv, affine._handle, wd._handle,
            h(cbias.data.compact()) if cbias is not None else None, out._handle, *geo, part_out._handle,
            _BN_PARTIAL_ROWS)
        if not done:
            return None
        node = cls.__new__(cls)
        node.x, node.gamma, node.beta, node.w, node.cbias = x, bn.weight, bn.bias, w, cbias
        node.xd, node.wd, node.gd = xd, wd, gd
        node.mean, node.var, node.affine, node.eps, node.geo = mean, var, affine, float(bn.eps), geo
        node.bn_partials = (part_out, rows) if rows > 0 else None
        node._grads = None
        _FusedNode.__init__(node, out, [x, bn.weight, bn.bias, w, cbias])
        return node

    def _pending_ok(self):
        """the convolution's input- and weight-gradient kernels can take a pending BatchNorm backward as their dY"""
        mod = self.device.mod
        if self.cbias is not None:
            return False
        dg, wg, _ = mod.conv2d_bn_bwd_supported(*self.geo)
        return dg and (wg or not self.w.requires_grad)

    _defer_in = None

    def _defer_bn(self):
        """this node's own BatchNorm (in front of the convolution) hands ITS backward pass to the node that made x"""
        if self._defer_in is None:
            self._defer_in = bool(self.x.requires_grad and self.gamma is not None and _accepts_pending(self.x)
                                  and self.device.mod.conv2d_bn_bwd_supported(*self.geo)[0])
        return self._defer_in

    def grad_fn(self, node, grad):
        if self._grads is None or self._grads[0] is not grad:
            pend = getattr(grad, "bn_pending", None)
            g = grad.compact()
            dev = g.device
            N, C, H, W, O, k, stride, pad = self.geo
            h = lambda t: t._handle if t is not None else None
            dw = dcb = dx = dg = db = None
            bn_needs = (self.x.requires_grad or (self.gamma is not None and self.gamma.requires_grad)
                        or (self.beta is not None and self.beta.requires_grad))
            gp = None                                          # the output gradient written out, where that is needed
            da = None
            if bn_needs:
                da = _new(self.xd.shape, dev)                  # gradient of the (never materialised) normalised tensor
                defer = _BN_BWD_FUSED and self._defer_bn()
                if pend is not None or defer:
                    # (the input gradient first: its launch writes the table the weight gradient reads)
                    targets = [(self.xd._handle, self.mean._handle)] if defer else None
                    cap = dev.mod.conv2d_bn_bwd_supported(*self.geo)[2]
                    rows = _new_rows(cap, C, 2, dev) if defer else None
                    nrows = dev.conv2d_dgrad_bn(g._handle, self.wd._handle, None, da._handle, *self.geo, pend, targets,
                                                h(rows), cap)
                    if pend is not None:
                        pend.table_ready = True
                    dg, db = _grad_out(self.gamma, (1, C, 1, 1), dev), _grad_out(self.beta, (1, C, 1, 1), dev)
                    if defer:
                        table = _new((C, 4), dev)
                        mine = _PendingBN(self.xd._handle, rows._handle, nrows, cap, 2, 1, self.mean._handle, self.var._handle,
                                          h(self.gd), self.eps, (N, C, H * W), dg._handle, db._handle, table._handle,
                                          (self.xd, rows, self.mean, self.var, self.gd, dg, db, table, da))
                        dx = _alias(da)
                        dx.bn_pending = mine
                    else:
                        dx = _new(self.xd.shape, dev) if self.x.requires_grad else None
                        dev.bn2d_backward(da._handle, self.xd._handle, h(self.gd), self.mean._handle, self.var._handle,
                                          h(dx), dg._handle, db._handle, N, C, H * W, self.eps)
                else:
                    dev.conv2d_dgrad(g._handle, self.wd._handle, da._handle, *self.geo)
                    dx = _new(self.xd.shape, dev) if self.x.requires_grad else None
                    dg, db = _grad_out(self.gamma, (1, C, 1, 1), dev), _grad_out(self.beta, (1, C, 1, 1), dev)
                    dev.bn2d_backward(da._handle, self.xd._handle, h(self.gd), self.mean._handle, self.var._handle,
                                      h(dx), dg._handle, db._handle, N, C, H * W, self.eps)
            if self.w.requires_grad:
                dw = _grad_out(self.w, self.wd.shape, dev)
                queued = False
                if pend is not None:
                    _pending_table(pend, dev)
                    queued = dev.conv2d_wgrad_bn(self.xd._handle, self.affine._handle, g._handle, pend.z, pend.table,
                                                 dw._handle, *self.geo)
                    if not queued:
                        gp = _pending_materialise(pend, g)
                if not queued:
                    dev.conv2d_wgrad_affine(self.xd._handle, self.affine._handle, (gp if gp is not None else g)._handle, dw._handle, *self.geo)
            if self.cbias is not None and self.cbias.requires_grad:
                if pend is not None and gp is None:
                    gp = _pending_materialise(pend, g)
                gb = gp if gp is not None else g
                dcb = _grad_out(self.cbias, self.cbias.shape, dev)
                dev.reduce_to_shape(gb._handle, dcb._handle, gb.shape, self.cbias.shape)
            self._grads = (grad, dx, dg, db, dw, dcb)
        _, dx, dg, db, dw, dcb = self._grads
        if node is self.x:
            return dx
        if node is self.gamma:
            return dg
        if node is self.beta:
            return db
        if node is self.w:
            return dw
        return dcb




def bn_conv2d_train(x, bn, conv):
    """conv(bn(x)) for a training-mode BatchNorm2d `bn` directly in front of the Conv2d `conv`, with the normalisation
    applied on the convolution's operand loads (one node, no normalised tensor).  Returns None when that form does
    not apply (not a fused device, evaluation mode, x carries no convolution partials, or the shape has no such
    kernel): the caller then writes conv(bn(x))."""
    x = realize(x)
    mod = x.device.mod
    if not (_fused(x) and hasattr(mod, "conv2d_forward_bn_input") and bn.training
            and is_grad_enable() and bn.weight is not None and bn.bias is not None):
        return None
    return _BnConv2dFused.build(x, bn, conv)


def batch_norm2d_pair_fusable(xa, bn_a, xb, bn_b):
    """True when bn_a(xa) + bn_b(xb) can run as the paired node: fused device, both layers in training
    mode with affine parameters, equal shapes, both inputs carry convolution partials."""
    xa, xb = realize(xa), realize(xb)
    mod = xa.device.mod
    return (_fused(xa) and hasattr(mod, "bn2d_forward_partials_pair") and bn_a.training and bn_b.training
            and is_grad_enable() and tuple(xa.shape) == tuple(xb.shape) and len(xa.shape) == 4
            and getattr(xa, "bn_partials", None) is not None and getattr(xb, "bn_partials", None) is not None
            and bn_a.weight is not None and bn_b.weight is not None and xa.device == xb.device)


def batch_norm2d_pair_train(xa, bn_a, xb, bn_b):
    xa, xb = realize(xa), realize(xb)
    return _BatchNorm2dPairFused(xa, bn_a, xb, bn_b)


class _BnReluPoolFused(_FusedNode):
    """maxpool2d(relu(batch_norm2d_train(x)), 2) as one node: one forward launch after the
    statistics pass, two backward launches; the intermediate full-resolution tensors of the three
    operators are never written (they are recomputed from x in the backward)."""

    def __init__(self, x, weight, bias, running_mean, running_var, eps, momentum):
        self.x, self.weight, self.bias = x, weight, bias
        N, C, H, W = (int(v) for v in x.shape)
        self.geo = (N, C, H, W)
        self.eps = float(eps)
        dev = x.device
        self.xd = x.data.compact()
        self.wd = weight.data.compact() if weight is not None else None
        self.bd = bias.data.compact() if bias is not None else None
        out = _new((N, C, H // 2, W // 2), dev)
        self.mean, self.var = _new((1, C, 1, 1), dev), _new((1, C, 1, 1), dev)
        h = lambda t: t._handle if t is not None else None
        rm = h(running_mean.data) if running_mean is not None else None
        rv = h(running_var.data) if running_var is not None else None
        partials = getattr(x, "bn_partials", None)
        if partials is not None and hasattr(dev.mod, "bn_relu_maxpool2_forward_partials"):
            # x is a convolution output that carries BatchNorm partials: no statistics pass over it
            dev.bn_relu_maxpool2_forward_partials(self.xd._handle, partials[0]._handle, partials[1], h(self.wd),
                                                  h(self.bd), out._handle, self.mean._handle, self.var._handle,
                                                  rm, rv, N, C, H, W, self.eps, float(momentum))
        else:
            dev.bn_relu_maxpool2_forward(self.xd._handle, h(self.wd), h(self.bd), out._handle, self.mean._handle,
                                         self.var._handle, rm, rv, N, C, H, W, self.eps, float(momentum))
        self._grads = None
        super().__init__(out, [x, weight, bias])

    def grad_fn(self, node, grad):
        if self._grads is None or self._grads[0] is not grad:
            dev = grad.device
            N, C, H, W = self.geo
            dg, db = _grad_out(self.weight, (1, C, 1, 1), dev), _grad_out(self.bias, (1, C, 1, 1), dev)
            h = lambda t: t._handle if t is not None else None
            conv = self.x if isinstance(self.x, _Conv2dFused) else None
            if (conv is not None and _STEM_FUSED_BWD and hasattr(dev.mod, "stem_backward") and self.x.requires_grad
                    and not conv.x.requires_grad and conv.bias is None and conv.w.requires_grad
                    and conv._dw_ready is None and len(conv.children) == 1 and conv.children[0] is self):
                # (only when this node is the convolution output's ONLY consumer: another consumer's contribution
                # would have to be added to a gradient tensor that is never formed)
                # the convolution in front needs only its weight gradient: one statistics launch, then the weight
                # gradient rebuilds d loss / d x from x while it stages it -- that tensor is never written.  The
                # convolution's node receives a placeholder as its output gradient and hands the stored dw out.
                dw = _grad_out(conv.w, conv.wd.shape, dev)
                if dev.stem_backward(conv.xd._handle, self.xd._handle, self.data._handle, grad.compact()._handle,
                                     h(self.wd), h(self.bd), self.mean._handle, self.var._handle, dw._handle,
                                     dg._handle, db._handle, *conv.geo, self.eps):
                    placeholder = _new(self.xd.shape, dev)      # (a pool block: nothing is written to it)
                    conv._dw_ready = (dw, placeholder)
                    self._grads = (grad, placeholder, dg, db)
            if self._grads is None or self._grads[0] is not grad:
                dx = _new(self.xd.shape, dev) if self.x.requires_grad else None
                dev.bn_relu_maxpool2_backward(self.xd._handle, self.data._handle, grad.compact()._handle, h(self.wd),
                                              h(self.bd), self.mean._handle, self.var._handle, h(dx), dg._handle,
                                              db._handle, N, C, H, W, self.eps)
                self._grads = (grad, dx, dg, db)
        _, dx, dg, db = self._grads
        if node is self.x:
            return dx
        return dg if node is self.weight else db




# (test hook, not an environment switch: tests/test_hip_parity.py sets it to False to get every operator's own backward launch
# as the comparison path of the fused stem backward)
_STEM_FUSED_BWD = True


def bn_relu_maxpool2_fusable(x, bn, pool):
    """True when pool(relu(bn(x))) can run as the fused stem node: fused device, training mode,
    2x2 / stride-2 / unpadded pooling, even height, width a multiple of 8."""
    x = realize(x)
    if not (_fused(x) and bn.training and hasattr(x.device.mod, "bn_relu_maxpool2_forward")):
        return False
    k = getattr(pool, "kernel_size", None)
    st = getattr(pool, "stride", k)
    if k != 2 or (st not in (None, 0, 2)) or getattr(pool, "padding", 0) not in (0, None):
        return False
    return len(x.shape) == 4 and int(x.shape[2]) % 2 == 0 and int(x.shape[3]) % 8 == 0


def bn_relu_maxpool2_train(x, bn):
    x = realize(x)
    return _BnReluPoolFused(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)


class _StemNoZFused(_FusedNode):
    """pool(relu(bn(conv(x)))) of the CIFAR stem (test/ResNet_CIFAR10_cuda.py:84-87) as ONE node whose
    full-resolution convolution output never exists in memory (dfhip_stem.hip): forward = the convolution keeping only
    the window extreme that survives bn + relu + max, then bn + relu on that; backward = the convolution rebuilt from x
    inside the statistics pass and inside the weight gradient.  Values equal the three operators' (the tie rules
    included); x must not need a gradient."""

    @classmethod
    def build(cls, x, conv, bn, pool):
        dev = x.device
        fn = getattr(dev.mod, "stem_forward_noz", None)
        k = getattr(pool, "kernel_size", None)
        st = getattr(pool, "stride", k)
        if (fn is None or not _fused(x) or not bn.training or not is_grad_enable() or x.requires_grad
                or len(x.shape) != 4 or bn.weight is None or bn.bias is None
                or k != 2 or (st not in (None, 0, 2)) or getattr(pool, "padding", 0) not in (0, None)):
            return None
        N, C, H, W = (int(v) for v in x.shape)
        O, _, ks, _ = (int(v) for v in conv.weight.shape)
        geo = (N, C, H, W, O, ks, int(conv.stride), int(conv.padding))
        xd, wd = x.data.compact(), conv.weight.data.compact()
        cb = conv.bias.data.compact() if conv.bias is not None else None     # (1, O, 1, 1): conv.py:104-108 adds it
        gd, bd = bn.weight.data.compact(), bn.bias.data.compact()
        out = _new((N, O, H // 2, W // 2), dev)
        mean, var = _new((1, O, 1, 1), dev), _new((1, O, 1, 1), dev)
        rm = bn.running_mean.data._handle if bn.running_mean is not None else None
        rv = bn.running_var.data._handle if bn.running_var is not None else None
        if not fn(xd._handle, wd._handle, cb._handle if cb is not None else None, gd._handle, bd._handle, out._handle,
                  mean._handle, var._handle, rm, rv, *geo, float(bn.eps), float(bn.momentum)):
            return None
        node = cls.__new__(cls)
        node.w, node.cbias, node.weight, node.bias = conv.weight, conv.bias, bn.weight, bn.bias
        node.geo, node.eps = geo, float(bn.eps)
        node._saved = (xd, wd, cb, gd, bd, mean, var)
        node.mean, node.var = mean, var
        node._grads = None
        _FusedNode.__init__(node, out, [conv.weight, conv.bias, bn.weight, bn.bias])
        return node

    def grad_fn(self, node, grad):
        if self._grads is None or self._grads[0] is not grad:
            dev = grad.device
            xd, wd, cb, gd, bd, mean, var = self._saved
            O = self.geo[4]
            dw, dg, db = _grad_out(self.w, wd.shape, dev), _grad_out(self.weight, (1, O, 1, 1), dev), _grad_out(self.bias, (1, O, 1, 1), dev)
            dcb = _grad_out(self.cbias, (1, O, 1, 1), dev) if cb is not None else None
            h = lambda t: t._handle if t is not None else None
            if not dev.stem_backward_noz(xd._handle, wd._handle, h(cb), self.data._handle, grad.compact()._handle,
                                         gd._handle, bd._handle, mean._handle, var._handle, dw._handle, h(dcb),
                                         dg._handle, db._handle, *self.geo, self.eps):
                raise RuntimeError("stem backward: the shape the forward kernel took has no backward kernel")
            self._grads = (grad, dw, dcb, dg, db)
        _, dw, dcb, dg, db = self._grads
        if node is self.w:
            return dw
        if self.cbias is not None and node is self.cbias:
            return dcb
        return dg if node is self.weight else db




def stem_conv_bn_relu_pool_train(x, conv, bn, pool):
    """pool(relu(bn(conv(x)))) as one node without the convolution's output tensor, or None when that form does not
    apply (the caller then writes the three calls)."""
    x = realize(x)
    return _StemNoZFused.build(x, conv, bn, pool)


class _BatchNorm2dEvalFused(_FusedNode):
    def __init__(self, x, weight, bias, running_mean, running_var, eps):
        N, C, H, W = (int(v) for v in x.shape)
        out = _new(x.shape, x.device)
        h = lambda t: t.data.compact()._handle if t is not None else None
        x.device.bn2d_forward_eval(x.data.compact()._handle, h(weight), h(bias), out._handle,
                                   running_mean.data._handle, running_var.data._handle, N, C, H * W,
                                   float(eps))
        super().__init__(out, [])


def batch_norm2d_eval(x, weight, bias, running_mean, running_var, eps):
    x = realize(x)
    return _BatchNorm2dEvalFused(x, weight, bias, running_mean, running_var, eps)
