"""GPU: Dropout, the classifier behind it and the loss from ONE launch (dfhip_droplin_ce_forward + dfhip_head_grads;
nn/lazy.py "drop" / "droplin").

Reference: test/CNN_CIFAR10_cuda.py:104-107 (`x = self.drop(x); x = self.fc(x)`) and its CrossEntropyLoss =
nn/modules/dropout.py:12-33 (mask from np.random.binomial on the host, x * mask / (1 - p)), nn/functional.py:8-12, 104-115 and
their backward passes through autograd (tensor.py:640-676, 699-716).  The oracle runs those primitive compositions on the
NumPy device with the same seed (= the same mask); the hip device launches one kernel for all of it in the forward pass and
only the sums over the rows (dW, db) in the backward pass."""
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_close

sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _run(dev, x_np, y_np, w_np, b_np, p, scenario, conv_w=None):
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor, realize
    np.random.seed(17)                                       # the module draws its mask on the host: the same on both devices
    x = Tensor(x_np, device=dev, requires_grad=True)
    fc = nn.Linear(w_np.shape[0], w_np.shape[1], device=dev)
    fc.weight.data = Tensor(w_np, device=dev).data
    fc.bias.data = Tensor(b_np, device=dev).data
    drop = nn.Dropout(p)
    drop.train()
    conv = None
    h = x
    if conv_w is not None:                                   # a convolution in front: its weight gradient opens a group
        conv = nn.Conv2d(conv_w.shape[1], conv_w.shape[0], kernel_size=3, padding=1, bias=False, device=dev)
        conv.weight.data = Tensor(conv_w, device=dev).data
        h = realize(conv(x))
    flat = h.reshape(h.shape[0], -1)
    hd = drop(flat)
    logits = fc(hd)
    loss = nn.CrossEntropyLoss()(logits, Tensor(y_np, device=dev))
    if scenario == "logits_used_twice":                      # something else differentiates through the logits
        loss = loss + tensor.mean(logits * logits) * 0.25
    if scenario == "dropout_used_twice":                     # ... or reads Dropout's output (its record launches its own node)
        loss = loss + tensor.mean(hd * hd) * 0.5
    loss.backward()
    out = {"loss": loss.numpy().copy(), "logits": logits.numpy().copy(), "dx": x.grad.numpy().copy(),
           "dw": fc.weight.grad.numpy().copy(), "db": fc.bias.grad.numpy().copy()}
    if conv is not None:
        out["dconv"] = conv.weight.grad.numpy().copy()
    Graph.free_graph_all()
    return out


CASES = {
    # (x shape, P, p, convolution in front)
    "cnn_bs256_2048": ((256, 2048), 10, 0.5, False),
    "tail_in_a_group": ((256, 256, 2, 2), 10, 0.5, True),      # (256 -> 256 on 2 x 2 images: a grouped weight gradient)
    "ragged_bs37_300": ((37, 300), 7, 0.3, False),
    "c64_p16": ((5, 64), 16, 0.8, False),
    "c1000_bs1200": ((1200, 1000), 10, 0.5, False),
    "c512": ((64, 512), 10, 0.5, False),
    "odd_c301_p7": ((19, 301), 7, 0.5, False),                # (weights that are no whole 16-byte pieces)
    "c2048_p16": ((48, 2048), 16, 0.25, False),               # (the largest weight tile: 128 KB of LDS)
}


@pytest.mark.parametrize("scenario", ["plain", "logits_used_twice", "dropout_used_twice"])
@pytest.mark.parametrize("name", list(CASES))
def test_dropout_classifier_and_loss_from_one_launch(hip_device, cpu_oracle, name, scenario):
    shape, P, p, with_conv = CASES[name]
    rng = np.random.default_rng(len(name) + shape[0])
    x = rng.standard_normal(shape).astype(np.float32)
    N = shape[0]
    C = int(np.prod(shape[1:]))
    y = np.eye(P, dtype=np.float32)[rng.integers(0, P, N)] * 0.95 + 0.05 / P
    w = (rng.standard_normal((C, P)) / np.sqrt(C)).astype(np.float32)
    b = rng.standard_normal((1, P)).astype(np.float32)
    conv_w = (rng.standard_normal((shape[1], shape[1], 3, 3)) / np.sqrt(9 * shape[1])).astype(np.float32) if with_conv else None
    mod = hip_device.mod
    mod.prof_reset(); mod.prof_enable(True)
    got = _run(hip_device, x, y, w, b, p, scenario, conv_w)
    mod.synchronize(); mod.prof_enable(False)
    fams = {k: v["calls"] for k, v in mod.prof_read().items() if v["calls"]}
    want = _run(cpu_oracle, x, y, w, b, p, scenario, conv_w)
    for k in want:
        assert_close(got[k], want[k], 1e-4, f"{name} {scenario} {k}")
    if not with_conv:
        assert (got["dx"] == 0).any() and (got["dx"] != 0).any()          # (the gradient at Dropout's input carries the mask)
    assert fams.get("droplin_ce_fwd_bwd") == 1, fams
    assert "softmax_ce_fwd" not in fams and "gemm_mfma_f32" not in fams, fams
    if scenario == "plain":
        assert "dropout_fwd" not in fams and "dropout_bwd" not in fams and "pool_linear_bwd" not in fams, fams
        if with_conv:
            assert "head_grads" not in fams, fams                # extra workgroups of the group's slab sum
        else:
            assert fams.get("head_grads") == 1, fams
    elif scenario == "logits_used_twice":
        assert fams.get("pool_linear_bwd") == 1 and fams.get("dropout_bwd") == 1, fams
    else:
        assert fams.get("dropout_fwd") == 1 and fams.get("dropout_bwd") == 1, fams       # the record's own node, same mask


def test_fused_tail_has_the_unfused_values(hip_device):
    """Dropout's output, the logits' expressions and the loss's order of additions are the separate nodes' (DF_DROPLIN_CE=0):
    the same loss bits; logits and gradients within rounding of another summation order"""
    from deepflows_amd.nn import lazy
    rng = np.random.default_rng(5)
    N, C, P = 256, 2048, 10
    x = rng.standard_normal((N, C)).astype(np.float32)
    y = np.eye(P, dtype=np.float32)[rng.integers(0, P, N)]
    w = (rng.standard_normal((C, P)) / np.sqrt(C)).astype(np.float32)
    b = rng.standard_normal((1, P)).astype(np.float32)
    outs = []
    for on in (True, False):
        lazy.DROPLIN_CE = on
        try:
            outs.append(_run(hip_device, x, y, w, b, 0.5, "plain"))
        finally:
            lazy.DROPLIN_CE = True
    for k in ("loss", "logits", "dx", "dw", "db"):
        assert_close(outs[0][k], outs[1][k], 2e-5, k)
    assert np.array_equal(outs[0]["dx"] == 0, outs[1]["dx"] == 0)       # the same mask


def test_evaluation_mode_and_no_grad_take_the_ordinary_path(hip_device, cpu_oracle):
    from deepflows_amd import nn
    from deepflows_amd.autograd import no_grad
    from deepflows_amd.tensor import Graph, Tensor
    rng = np.random.default_rng(8)
    x_np = rng.standard_normal((32, 256)).astype(np.float32)
    y_np = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 32)]
    res = {}
    for name, dev in (("hip", hip_device), ("cpu", cpu_oracle)):
        np.random.seed(3)
        fc = nn.Linear(256, 10, device=dev)
        drop = nn.Dropout(0.5)
        drop.eval()
        a = nn.CrossEntropyLoss()(fc(drop(Tensor(x_np, device=dev, requires_grad=True))), Tensor(y_np, device=dev)).numpy().copy()
        drop.train()
        np.random.seed(4)
        with no_grad():
            c = nn.CrossEntropyLoss()(fc(drop(Tensor(x_np, device=dev))), Tensor(y_np, device=dev)).numpy().copy()
        res[name] = (a, c)
        Graph.free_graph_all()
    assert_close(res["hip"][0], res["cpu"][0], 1e-4, "eval")
    assert_close(res["hip"][1], res["cpu"][1], 1e-4, "no_grad")
