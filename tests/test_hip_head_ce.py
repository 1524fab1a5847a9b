"""GPU: the classifier head and its loss from ONE launch (dfhip_head_ce_forward + dfhip_head_grads; nn/lazy.py "head").

Reference: test/ResNet_CIFAR10_cuda.py:96-99 (`tensor.mean(x, axis=2)` twice, `self.fc(x)`) + :186 (`criterion(outputs,
targets)` = nn/functional.py:104-115) and their backward passes (tensor.py:699-716, 753-766; nn/functional.py:104-115 through
autograd).  The oracle runs exactly those primitive compositions on the NumPy device; the hip device launches one kernel for
all of it in the forward pass and only the sums over the images (dW, db) in the backward pass -- inside the sweep's grouped
slab sum when a convolution's weight gradient is queued, by themselves otherwise."""
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_close

sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _run(dev, x_np, y_np, w_np, b_np, scenario, conv_w=None):
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor, realize
    x = Tensor(x_np, device=dev, requires_grad=True)
    fc = nn.Linear(w_np.shape[0], w_np.shape[1], device=dev)
    fc.weight.data = Tensor(w_np, device=dev).data
    fc.bias.data = Tensor(b_np, device=dev).data
    conv = None
    h = x
    if conv_w is not None:                                   # a convolution in front: its weight gradient opens a group
        conv = nn.Conv2d(conv_w.shape[1], conv_w.shape[0], kernel_size=3, padding=1, bias=False, device=dev)
        conv.weight.data = Tensor(conv_w, device=dev).data
        h = realize(conv(x))                                 # (a tensor that exists, as a residual block's output is)
    feat = tensor.mean(tensor.mean(h, axis=2), axis=2)
    logits = fc(feat)
    loss = nn.CrossEntropyLoss()(logits, Tensor(y_np, device=dev))
    extra = {}
    if scenario == "logits_used_twice":                      # something else differentiates through the logits
        loss = loss + tensor.mean(logits * logits) * 0.25
    loss.backward()
    out = {"loss": loss.numpy().copy(), "logits": logits.numpy().copy(), "dx": x.grad.numpy().copy(),
           "dw": fc.weight.grad.numpy().copy(), "db": fc.bias.grad.numpy().copy()}
    if conv is not None:
        out["dconv"] = conv.weight.grad.numpy().copy()
    out.update(extra)
    Graph.free_graph_all()
    return out


CASES = {
    # (N, C, H, W, P, convolution in front)
    "r18_bs256": (256, 256, 2, 2, 10, False),
    "r18_bs256_in_a_group": (256, 256, 2, 2, 10, True),
    "ragged_bs37": (37, 96, 2, 2, 10, False),
    "wide_c512_4x4": (64, 512, 4, 4, 10, False),
    "hw3_p16": (40, 64, 1, 3, 16, False),
    "bs1500": (1500, 64, 2, 2, 7, False),
}


@pytest.mark.parametrize("scenario", ["plain", "logits_used_twice"])
@pytest.mark.parametrize("name", list(CASES))
def test_head_and_loss_from_one_launch(hip_device, cpu_oracle, name, scenario):
    N, C, H, W, P, with_conv = CASES[name]
    rng = np.random.default_rng(len(name) + N)
    cin = C                                                   # (256 -> 256 on 2 x 2 images: 64 workgroups, enough for a group)
    x = rng.standard_normal((N, cin, H, W)).astype(np.float32)
    y = np.eye(P, dtype=np.float32)[rng.integers(0, P, N)] * 0.95 + 0.05 / P
    w = (rng.standard_normal((C, P)) / np.sqrt(C)).astype(np.float32)
    b = rng.standard_normal((1, P)).astype(np.float32)
    conv_w = (rng.standard_normal((C, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32) if with_conv else None
    # (the reference's mean(axis) divides by the TOTAL element count, SURVEY F5: the pooled features are ~1e-6 -- scale the
    # classifier so that the logits are of order one and the softmax is not flat)
    w *= np.float32(float(N * C * H * W) * float(N * C * W) / (H * W))
    mod = hip_device.mod
    mod.prof_reset(); mod.prof_enable(True)
    got = _run(hip_device, x, y, w, b, scenario, conv_w)
    mod.synchronize(); mod.prof_enable(False)
    fams = {k: v["calls"] for k, v in mod.prof_read().items() if v["calls"]}
    want = _run(cpu_oracle, x, y, w, b, scenario, conv_w)
    for k in want:
        assert_close(got[k], want[k], 1e-4, f"{name} {scenario} {k}")
    # one launch for head + loss; the unfused nodes are not launched
    assert fams.get("head_ce_fwd_bwd") == 1, fams
    assert "pool_linear_fwd" not in fams and "softmax_ce_fwd" not in fams, fams
    if scenario == "plain":
        assert "pool_linear_bwd" not in fams, fams
        if with_conv:
            assert "head_grads" not in fams, fams                # extra workgroups of the group's slab sum
        else:
            assert fams.get("head_grads") == 1, fams
    else:
        assert fams.get("pool_linear_bwd") == 1, fams             # another gradient arrived: the ordinary backward launch


def test_head_loss_is_the_unfused_loss_bit_for_bit(hip_device):
    """the row losses are added up in dfhip_softmax_ce_forward's order; logits, d loss / d logits and the input gradient use
    the unfused kernels' expressions: the same bits as the three nodes (DF_HEAD_CE=0)"""
    from deepflows_amd.nn import lazy
    N, C, H, W, P = 256, 256, 2, 2, 10
    rng = np.random.default_rng(5)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    y = np.eye(P, dtype=np.float32)[rng.integers(0, P, N)]
    w = (rng.standard_normal((C, P)) / np.sqrt(C) * (float(N * C * H * W) * float(N * C * W) / (H * W))).astype(np.float32)
    b = rng.standard_normal((1, P)).astype(np.float32)
    outs = []
    for on in (True, False):
        lazy.HEAD_CE = on
        try:
            outs.append(_run(hip_device, x, y, w, b, "plain"))
        finally:
            lazy.HEAD_CE = True
    for k in ("loss", "logits", "dx"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    for k in ("dw", "db"):                                      # (their sums over the images run in another order)
        assert_close(outs[0][k], outs[1][k], 1e-5, k)


def test_a_loss_read_before_the_backward_pass_is_finished_by_one_small_launch(hip_device, cpu_oracle):
    """the head's launch leaves row losses; whoever reads the loss first gets their mean from dfhip_loss_finish (once), and
    the backward pass still delivers every gradient"""
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor
    N, C, P = 64, 128, 10
    rng = np.random.default_rng(9)
    x_np = rng.standard_normal((N, C, 2, 2)).astype(np.float32)
    y_np = np.eye(P, dtype=np.float32)[rng.integers(0, P, N)]
    w_np = (rng.standard_normal((C, P)) / np.sqrt(C) * (float(N * C * 4) * float(N * C * 2) / 4)).astype(np.float32)
    b_np = rng.standard_normal((1, P)).astype(np.float32)
    res = {}
    for name, dev in (("hip", hip_device), ("cpu", cpu_oracle)):
        x = Tensor(x_np, device=dev, requires_grad=True)
        fc = nn.Linear(C, P, device=dev)
        fc.weight.data = Tensor(w_np, device=dev).data
        fc.bias.data = Tensor(b_np, device=dev).data
        if name == "hip":
            dev.mod.prof_reset(); dev.mod.prof_enable(True)
        loss = nn.CrossEntropyLoss()(fc(tensor.mean(tensor.mean(x, axis=2), axis=2)), Tensor(y_np, device=dev))
        first = loss.numpy().copy()                     # read BEFORE backward
        again = loss.numpy().copy()
        loss.backward()
        after = loss.numpy().copy()
        if name == "hip":
            dev.mod.synchronize(); dev.mod.prof_enable(False)
            fams = {k: v["calls"] for k, v in dev.mod.prof_read().items() if v["calls"]}
            assert fams.get("loss_finish") == 1 and fams.get("head_ce_fwd_bwd") == 1 and fams.get("head_grads") == 1, fams
        assert np.array_equal(first, again) and np.array_equal(first, after)
        res[name] = (first, x.grad.numpy().copy(), fc.weight.grad.numpy().copy(), fc.bias.grad.numpy().copy())
        Graph.free_graph_all()
    for a, b, what in zip(res["hip"], res["cpu"], ("loss", "dx", "dw", "db")):
        assert_close(a, b, 1e-4, what)
