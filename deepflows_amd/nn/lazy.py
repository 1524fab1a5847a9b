"""Deferred module calls: how a model written as the reference writes it reaches the fused kernels.

The reference's scripts call one module after the other (test/ResNet_CIFAR10_cuda.py:31-47, 84-99):

    out = self.conv1(x); out = self.bn1(out); out = self.conv2(out); out = self.bn2(out)      # a residual block
    for layer in self.downsample: identity = layer(identity)
    out = out + identity
    x = self.conv1(x); x = self.bn1(x); x = self.relu(x); x = self.pool1(x)                   # the stem
    x = tensor.mean(x, axis=2); x = tensor.mean(x, axis=2); x = self.fc(x)                    # the head

The hip device has kernels for the SEQUENCES (the stem without its full-resolution tensor; BatchNorm applied on the next
convolution's operand loads; a block's 3x3 / stride-2 convolution and its 1x1 shortcut from one launch; the block's two
BatchNorms and its sum in one pass; the two means and the Linear in one launch).  `deepflows_amd/workloads.py` used to reach
them through a forward() rewritten around private entry points.  Here the modules themselves do it: on a fused device, in
training mode, with gradients enabled, Conv2d / BatchNorm2d / ReLU return a `tensor.LazyTensor` that only RECORDS the call
when its input is a recorded call it can extend, and the call that completes a sequence (MaxPool2d, `+`, Linear) -- or
anything else that touches the object -- launches.  What is launched for a sequence is exactly what the hand-written forward
launched; what is launched when a deferred result is touched on its own is the ordinary node of that module.

Recorded forms (x0 a real tensor):
    conv(x0)                                   "conv"
    bn(conv(..))                               "bn"        (training mode only)
    conv(bn(conv(..)))                         "conv" over "bn"
    relu(bn(conv(x0)))                         "relu"
    mean(x0, axis=2), mean(mean(x0, 2), 2)     "mean2", "meanhw"
    fc(x0), x0 two-dimensional                 "linear"
    fc(mean(mean(x0, 2), 2))                   "head"      (the classifier head; launched as one node when touched)
    relu(fc(x0)), fc2(relu(fc1(x0)))           "linrelu", "mlp2"   (test/MLP_MNIST_cuda.py:51-54)
    drop(x0), fc(drop(x0)), x0 two-dimensional "drop", "droplin"   (test/CNN_CIFAR10_cuda.py:104-107)
Completing calls: MaxPool2d(relu(bn(conv(x0)))), bn(conv(bn(conv(x0)))) + identity, F.relu(fc(x0)),
CrossEntropyLoss(head), CrossEntropyLoss(fc(drop(x0))) (round 6: head + loss + the per-image half of their backward pass
from one launch).
A record is single-shot: when a fused launch has run the BatchNorm of a record it did not hand out (`bn1` applied on conv2's
loads, `bn2` inside the block's tail, the stem's BatchNorm), whoever still holds the record and touches it later gets the same
values from a normalisation that does NOT update running_mean / running_var a second time (the reference calls each module
once: nn/modules/batchnorm.py:44-52), and such a record is not fused again (tests/test_hip_lazy_records.py).
DF_LAZY_MODULES=0 switches the mechanism off (every module call launches its own node, as before)."""
import os
import weakref

from .. import tensor
from ..autograd import is_grad_enable
from ..tensor import LazyTensor, Tensor, realize
from . import functional as F

ENABLED = os.environ.get("DF_LAZY_MODULES", "1") != "0"
MLP_CE = os.environ.get("DF_MLP_CE", "1") != "0"        # fc2(relu(fc1(x))) + CrossEntropyLoss: one launch each way (0: five launches)
HEAD_CE = os.environ.get("DF_HEAD_CE", "1") != "0"      # head + CrossEntropyLoss from one launch (0: the three nodes of rounds 1-5)
DROPLIN_CE = os.environ.get("DF_DROPLIN_CE", "1") != "0"   # Dropout + classifier + CrossEntropyLoss from one launch (0: three nodes)


def _state(x):
    """(op, input, module, extra) of a deferred result that has not been forced, else None"""
    if type(x) is not LazyTensor:
        return None
    st = object.__getattribute__(x, "_lz")
    return None if st[4] is not None else st


def _is(x, op):
    st = _state(x)
    return st is not None and st[0] == op


_STATS_DONE = "running statistics already updated"


def _consumed(bn_record):
    """A fused launch has just run the BatchNorm of this record (running_mean / running_var updated ONCE, as the reference's
    single module call does, nn/modules/batchnorm.py:44-52) without producing the record's own tensor.  Whoever still holds
    the record and touches it later gets the same values from a normalisation that leaves the running statistics alone."""
    st = _state(bn_record)
    if st is not None and st[0] == "bn":
        st[3] = _STATS_DONE


def _stats_done(bn_record):
    st = _state(bn_record)
    return st is not None and st[3] == _STATS_DONE


def _can_defer(x):
    """deferral applies to tensors of a fused device while gradients are recorded"""
    if not ENABLED or not is_grad_enable():
        return False
    if _state(x) is not None:
        return True
    x = realize(x)
    return isinstance(x, Tensor) and F._fused(x)


# ---- what the modules call ---------------------------------------------------------------------------------------------
def conv2d(module, x):
    """Conv2d.forward"""
    if _can_defer(x) and (_state(x) is None or _is(x, "bn")):
        return LazyTensor("conv", x, module)
    return F.conv2d(realize(x), module.weight, module.padding, module.stride, bias=module.bias)


def batch_norm2d(module, x):
    """BatchNorm2d.forward without a residual: deferred behind a deferred convolution in training mode; None = not
    deferred (the module runs its own forward on realize(x))"""
    if module.training and _is(x, "conv") and ENABLED and is_grad_enable():
        return LazyTensor("bn", x, module)
    return None


def _input(x):
    return _state(x)[1]


def _conv_of_real(x):
    """x is a recorded convolution of a tensor that exists"""
    return _is(x, "conv") and _state(_input(x)) is None


def relu(x):
    """F.relu / ReLU.forward on a recorded call: relu(bn(conv(x0))) stays recorded (a pooling layer may complete the stem);
    relu(fc(x0)) is one node (test/MLP_MNIST_cuda.py:51-52 `F.relu(self.fc1(x))`); None = not such a form"""
    if _is(x, "bn") and _conv_of_real(_input(x)):
        return LazyTensor("relu", x)
    if _is(x, "linear"):
        st = _state(x)
        x0 = realize(st[1])
        if MLP_CE and hasattr(x0.device.mod, "mlp2_ce_forward") and not x0.requires_grad:
            # (a second Linear and a CrossEntropyLoss may complete it: the whole perceptron's step as one launch each way)
            return LazyTensor("linrelu", x)
        # (the recorded fc(x) itself stays recorded: whoever still holds it and touches it later launches the plain Linear)
        return F.linear_relu(x0, st[2].weight, st[2].bias)
    return None


def max_pool2d(module, x):
    """MaxPool2d.forward: pool(relu(bn(conv(x0)))) is the stem; None = not that form"""
    if not _is(x, "relu"):
        return None
    bn_l = _state(x)[1]
    conv_l = _state(bn_l)[1]
    conv, bn = _state(conv_l)[2], _state(bn_l)[2]
    x0 = realize(_state(conv_l)[1])
    if _stats_done(bn_l):                                  # (this BatchNorm call has been launched by another consumer)
        return None
    out = F.stem_conv_bn_relu_pool_train(x0, conv, bn, module)
    if out is not None:                                    # one node, the convolution's output never exists
        _consumed(bn_l)
        return out
    z = realize(conv_l)
    if F.bn_relu_maxpool2_fusable(z, bn, module):
        out = F.bn_relu_maxpool2_train(z, bn)              # pool(relu(bn(z))) as one node
        _consumed(bn_l)
        return out
    return None                                            # (the caller pools realize(x))


def dropout(module, x, mask, keep):
    """Dropout.forward in training mode with its mask drawn: on a two-dimensional tensor of a fused device the call is recorded
    (a classifier-sized Linear and a CrossEntropyLoss may complete it); None = the module launches F.dropout_train"""
    if not (ENABLED and DROPLIN_CE and is_grad_enable() and _state(x) is None):
        return None
    xr = realize(x)
    if (isinstance(xr, Tensor) and F._fused(xr) and len(xr.shape) == 2 and xr.requires_grad and int(xr.shape[1]) <= 2048
            and hasattr(xr.device.mod, "droplin_ce_forward")):
        return LazyTensor("drop", xr, module, (mask, keep))
    return None


def linear(module, x):
    """Linear.forward: on mean(mean(x0, 2), 2) the classifier head as one node; on a 2-D tensor of a fused device the call is
    recorded (an F.relu may follow: one node for both); None = the module launches F.linear"""
    if _is(x, "drop") and int(module.weight.shape[1]) <= 16 and int(module.weight.shape[0]) == int(_state(x)[1].shape[1]):
        return LazyTensor("droplin", x, module)            # fc(drop(x0)), nothing of it launched yet
    if _is(x, "linrelu") and _state(_state(x)[1]) is not None:
        return LazyTensor("mlp2", x, module)               # fc2(relu(fc1(x0))), nothing of it launched yet
    if _is(x, "meanhw"):
        x0 = realize(_state(x)[1])
        if HEAD_CE and F._fused(x0) and hasattr(x0.device.mod, "head_ce_forward") and int(module.weight.shape[1]) <= 16:
            return LazyTensor("head", x0, module)          # (a CrossEntropyLoss may complete it: head + loss in one launch)
        return F.mean_hw_linear(x0, module)
    if ENABLED and is_grad_enable() and _state(x) is None:
        xr = realize(x)
        if isinstance(xr, Tensor) and F._fused(xr) and len(xr.shape) == 2 and hasattr(xr.device.mod, "linear_relu_forward"):
            return LazyTensor("linear", xr, module)
    return None


# ---- hooks of tensor.py ------------------------------------------------------------------------------------------------
def _mean(x, axis, keepdims):
    if not ENABLED or axis != 2 or keepdims or not is_grad_enable():
        return None
    if _is(x, "mean2"):
        return LazyTensor("meanhw", _state(x)[1])
    if _state(x) is None and isinstance(x, Tensor) and type(x) is not LazyTensor and F._fused(x) and len(x.shape) == 4 \
            and hasattr(x.device.mod, "reduce_axis"):
        return LazyTensor("mean2", x)
    return None


def _force(lz):
    """the node a deferred result stands for, launched on its own"""
    op, x, module, extra, _ = object.__getattribute__(lz, "_lz")
    if op == "conv":
        if _is(x, "bn") and not _stats_done(x):
            # conv(bn(z)): BatchNorm applied on the convolution's operand loads where the device has that form
            bn_l = _state(x)
            z = realize(bn_l[1])
            out = F.bn_conv2d_train(z, bn_l[2], module)
            if out is not None:
                _consumed(x)
                return out
        xr = realize(x)
        return F.conv2d(xr, module.weight, module.padding, module.stride, bias=module.bias)
    if op == "bn":
        if extra == _STATS_DONE:
            # a fused launch consumed this call: same batch statistics, same values, running statistics left alone (the
            # second node's gradients add up with the first's: the backward pass of a BatchNorm is linear in dy)
            return F.batch_norm2d_train(realize(x), module.weight, module.bias, None, None, module.eps, module.momentum)
        return module.forward(realize(x))
    if op == "linear":
        return F.linear(realize(x), module.weight, module.bias)
    if op == "head":
        return F.mean_hw_linear(realize(x), module)
    if op == "linrelu":
        lin = _state(x)
        if lin is not None:                                # relu(fc(x0)) as one node (test/MLP_MNIST_cuda.py:51-52)
            return F.linear_relu(realize(lin[1]), lin[2].weight, lin[2].bias)
        return F.relu(realize(x))                          # (somebody has launched the Linear meanwhile)
    if op == "mlp2":
        return F.linear(realize(x), module.weight, module.bias)
    if op == "drop":
        return F.dropout_train(realize(x), extra[0], extra[1])
    if op == "droplin":
        return F.linear(realize(x), module.weight, module.bias)
    if op == "relu":
        return F.relu(realize(x))
    if op == "mean2":
        return _plain_mean(realize(x))
    if op == "meanhw":
        return F.mean_hw(realize(x))
    raise RuntimeError(f"deferred call of unknown kind {op!r}")


def _plain_mean(x):
    node = object.__new__(tensor.mean)                     # (past mean.__new__: this call is not deferred again)
    node.__init__(x, axis=2)
    return node


def _add(a, b):
    """`a + b` with a deferred: bn2(conv2(..)) + identity is the tail of a residual block"""
    out = _block_tail(a, b)
    if out is not None:
        return out
    return tensor.add(realize(a), realize(b))


def _block_tail(out_l, identity):
    """test/ResNet_CIFAR10_cuda.py:31-47 from `out = self.bn2(out)` on: the launches of the hand-fused block
    (workloads.ResidualBlock.forward)"""
    if not _is(out_l, "bn") or not _is(_state(out_l)[1], "conv") or _stats_done(out_l):
        return None
    if _is(identity, "bn") and _stats_done(identity):
        return None
    bn2 = _state(out_l)[2]
    conv2_l = _state(out_l)[1]
    # the shortcut: bn_d(conv_d(x0)) not launched yet, or a tensor
    short = None
    if _is(identity, "bn") and _conv_of_real(_input(identity)):
        short = (_state(_input(identity))[2], _state(identity)[2], _input(identity))      # (conv_d, bn_d, conv_d's record)
    # conv1 of the block, when conv2 reads bn1(conv1(x0)) and nothing of it has been launched
    conv1_l = None
    c2_in = _state(conv2_l)[1]
    if _is(c2_in, "bn") and _conv_of_real(_input(c2_in)):
        conv1_l = _input(c2_in)
    zd = None
    if short is not None:
        conv_d, bn_d, convd_l = short
        xd0 = _state(convd_l)[1]
        if conv1_l is not None and _state(conv1_l)[1] is xd0:
            # conv1 and the shortcut convolution read the same tensor: one launch for both where the device has it
            conv1 = _state(conv1_l)[2]
            both = F._Conv2dFused.pair_s2(xd0, conv1, conv_d) if F._fused(xd0) else None
            if both is not None:
                z1, zd = both
                object.__getattribute__(conv1_l, "_lz")[4] = z1
                object.__getattribute__(convd_l, "_lz")[4] = zd
        if zd is None:
            zd = realize(convd_l)
        if conv1_l is not None:
            z1 = realize(conv1_l)
            if hasattr(z1, "shortcut_sibling") and hasattr(zd, "shortcut_sibling") and getattr(z1, "x", None) is getattr(zd, "x", 0):
                z1.shortcut_sibling = zd               # both read x: one launch forms both input gradients
                zd._sibling_host = weakref.ref(z1)
    z2 = realize(conv2_l)                              # conv2(bn1(z1)): bn1 on the operand loads where that form applies
    if short is not None:
        if F.batch_norm2d_pair_fusable(z2, bn2, zd, bn_d):
            out = F.batch_norm2d_pair_train(z2, bn2, zd, bn_d)         # bn2(z2) + bn_d(zd): one launch each way
            _consumed(out_l)
            _consumed(identity)
            return out
        # (realize(identity) below launches bn_d(zd) as the record's own node)
    out = bn2.forward(z2, residual=realize(identity))
    _consumed(out_l)
    return out


def head(x, module):
    """fc(mean(mean(x, 2), 2)) for a model that calls the fused head itself (workloads.ResNet18): recorded like the module
    calls, so that the criterion can complete it"""
    x = realize(x)
    if (ENABLED and HEAD_CE and is_grad_enable() and F._fused(x) and hasattr(x.device.mod, "head_ce_forward")
            and len(x.shape) == 4 and int(module.weight.shape[1]) <= 16):
        return LazyTensor("head", x, module)
    return F.mean_hw_linear(x, module)


def _head_ce(logits, target):
    """F.cross_entropy on a recorded classifier head: head + loss (+ the per-image half of their backward pass) from one
    launch; the record then holds the logits node.  None = not that form."""
    if _is(logits, "mlp2"):
        return _mlp2_ce(logits, target)
    if _is(logits, "droplin"):
        return _droplin_ce(logits, target)
    if not _is(logits, "head"):
        return None
    st = object.__getattribute__(logits, "_lz")
    tgt = realize(target)
    if not isinstance(tgt, Tensor):
        return None
    both = F.head_ce_train(st[1], st[2], tgt)
    if both is None:
        return None
    st[4] = both[0]
    return both[1]


def _droplin_ce(logits, target):
    """CrossEntropyLoss on fc(drop(x0)) with nothing of it launched: one launch; the record then holds the logits node (the
    Dropout's own record stays a record: whoever still holds it and touches it later launches a Dropout node with the same
    mask).  None = not that form."""
    st = object.__getattribute__(logits, "_lz")
    drop = _state(st[1])
    if drop is None:
        return None
    tgt = realize(target)
    if not isinstance(tgt, Tensor):
        return None
    both = F.droplin_ce_train(drop[1], drop[3][0], drop[3][1], st[2], tgt)
    if both is None:
        return None
    st[4] = both[0]
    return both[1]


def _mlp2_ce(logits, target):
    """CrossEntropyLoss on fc2(relu(fc1(x0))) with nothing of it launched: forward + the per-row half of the backward pass
    from one launch; the records then hold the nodes.  None = not that form."""
    st2 = object.__getattribute__(logits, "_lz")
    hid = st2[1]
    if not _is(hid, "linrelu"):
        return None
    sth = object.__getattribute__(hid, "_lz")
    lin = _state(sth[1])
    if lin is None:
        return None
    tgt = realize(target)
    if not isinstance(tgt, Tensor):
        return None
    nodes = F.mlp2_ce_train(lin[1], lin[2], st2[2], tgt)
    if nodes is None:
        return None
    sth[4], st2[4] = nodes[0], nodes[1]
    return nodes[2]


def _maximum0(x):
    """tensor.maximum(x, 0) on a recorded call or on a tensor of a fused device: F.relu's node (the reference's F.relu is this
    call, nn/functional.py:15-16); None = the primitive node"""
    if not ENABLED:
        return None
    if _state(x) is not None:
        return F.relu(x)
    xr = realize(x)
    if isinstance(xr, Tensor) and F._fused(xr):
        return F.relu(xr)
    return None


tensor._lazy_hooks.update({"force": _force, "add": _add, "mean": _mean, "maximum0": _maximum0})
F._lazy_relu = relu
F._lazy_head_ce = _head_ce
