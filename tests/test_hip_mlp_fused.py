"""GPU: the two-layer perceptron's training step as one launch each way (dfhip_mlp2_ce_forward / _backward; nn/lazy.py
"linrelu" / "mlp2").

Reference: test/MLP_MNIST_cuda.py:37-59 (fc1 -> F.relu -> fc2, CrossEntropyLoss, SGD lr 0.05) = nn/functional.py:8-16,
104-115 and tensor.py:699-716, 872-877, 462-483 through autograd.  The oracle runs those primitive compositions on the NumPy
device; BASELINE.json config 2 is the (784, 256, 10) model at batch 128."""
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_close

sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _steps(dev, sizes, x_np, y_np, steps, read_hidden=False, x_grad=False):
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Graph, Tensor
    np.random.seed(0)
    model = workloads.MLP(sizes, device=dev)
    model.train()
    opt = optim.SGD(model.parameters(), lr=0.05)
    crit = nn.CrossEntropyLoss()
    x, y = Tensor(x_np, device=dev, requires_grad=x_grad), Tensor(y_np, device=dev)
    out = {"loss": [], "logits": None}
    for _ in range(steps):
        if read_hidden:
            # the script written with the hidden activation kept and used twice (an auxiliary penalty)
            hdn = nn.functional.relu(model.fc1(x))
            logits = model.fc2(hdn)
            loss = crit(logits, y) + (hdn * hdn).mean() * 0.5
        else:
            logits = model(x)
            loss = crit(logits, y)
        opt.zero_grad()
        loss.backward()
        out["grads"] = [p.grad.numpy().copy() for p in model.parameters()]
        opt.step()
        out["loss"].append(float(loss.numpy()[0]))
        out["logits"] = logits.numpy().copy()
        if x_grad:
            out["dx"] = x.grad.numpy().copy()
            x.grad = None
        Graph.free_graph()
    out["params"] = [p.data.numpy().copy() for p in model.parameters()]
    Graph.free_graph_all()
    return out


CASES = {"mnist_784_256_10_bs128": ((784, 256, 10), 128), "ragged_48_64_7_bs37": ((48, 64, 7), 37),
         "hidden128_bs256": ((160, 128, 16), 256), "bs1030": ((32, 64, 10), 1030),
         # below 64 rows the forward pass is ONE launch (k_mlp2_fwd): every hidden width of it
         "one_launch_hidden128_bs48": ((64, 128, 10), 48), "one_launch_hidden256_bs40": ((784, 256, 10), 40)}


@pytest.mark.parametrize("name", list(CASES))
def test_mlp_step_is_one_launch_each_way(hip_device, cpu_oracle, name):
    sizes, M = CASES[name]
    rng = np.random.default_rng(len(name))
    x = rng.random((M, sizes[0]), dtype=np.float32)
    x[::3, ::5] = 0.0
    y = np.eye(sizes[-1], dtype=np.float32)[rng.integers(0, sizes[-1], M)]
    mod = hip_device.mod
    mod.prof_reset(); mod.prof_enable(True)
    got = _steps(hip_device, sizes, x, y, 3)
    mod.synchronize(); mod.prof_enable(False)
    fams = {k: v["calls"] for k, v in mod.prof_read().items() if v["calls"]}
    want = _steps(cpu_oracle, sizes, x, y, 3)
    assert_close(np.array(got["loss"]), np.array(want["loss"]), 1e-4, name + " losses")
    assert_close(got["logits"], want["logits"], 1e-4, name + " logits")
    for i, (a, b) in enumerate(zip(got["grads"], want["grads"])):
        assert_close(a, b, 1e-4, f"{name} grad {i}")
    for i, (a, b) in enumerate(zip(got["params"], want["params"])):
        assert_close(a, b, 1e-4, f"{name} param {i} after 3 SGD steps")
    assert fams.get("mlp2_ce_fwd") == 3 and fams.get("mlp2_ce_bwd") == 3, fams
    assert fams.get("mlp_fc1_fwd") == (3 if M >= 64 else None), fams      # (two launches forward from 64 rows on)
    assert "gemm_mfma_f32" not in fams and "softmax_ce_fwd" not in fams and "loss_finish" not in fams, fams


def test_relu_gradient_passes_at_zero_preactivation(hip_device, cpu_oracle):
    """tensor.py:872-877: maximum's gradient mask is (output == input): a pre-activation of exactly 0 passes the gradient"""
    from deepflows_amd import nn
    from deepflows_amd.tensor import Graph, Tensor
    M, K, HD, P = 8, 16, 64, 4
    rng = np.random.default_rng(2)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w1 = rng.standard_normal((K, HD)).astype(np.float32)
    w1[:, :8] = 0.0                                         # eight hidden units with x . w1 == 0 exactly ...
    b1 = rng.standard_normal((1, HD)).astype(np.float32)
    b1[:, :8] = 0.0                                         # ... and bias 0: pre-activation == 0
    w2 = rng.standard_normal((HD, P)).astype(np.float32)
    b2 = rng.standard_normal((1, P)).astype(np.float32)
    y = np.eye(P, dtype=np.float32)[rng.integers(0, P, M)]
    res = {}
    for name, dev in (("hip", hip_device), ("cpu", cpu_oracle)):
        fc1, fc2 = nn.Linear(K, HD, device=dev), nn.Linear(HD, P, device=dev)
        for t, v in ((fc1.weight, w1), (fc1.bias, b1), (fc2.weight, w2), (fc2.bias, b2)):
            t.data = Tensor(v, device=dev).data
        loss = nn.CrossEntropyLoss()(fc2(nn.functional.relu(fc1(Tensor(x, device=dev)))), Tensor(y, device=dev))
        loss.backward()
        res[name] = (loss.numpy().copy(), fc1.weight.grad.numpy().copy(), fc1.bias.grad.numpy().copy())
        Graph.free_graph_all()
    assert np.abs(res["cpu"][2][0, :8]).max() > 0          # (the oracle does pass the gradient there)
    for a, b, what in zip(res["hip"], res["cpu"], ("loss", "dW1", "db1")):
        assert_close(a, b, 1e-4, what)


def test_hidden_activation_used_twice_takes_the_ordinary_backward(hip_device, cpu_oracle):
    sizes, M = (48, 64, 7), 24
    rng = np.random.default_rng(4)
    x = rng.random((M, sizes[0]), dtype=np.float32)
    y = np.eye(sizes[-1], dtype=np.float32)[rng.integers(0, sizes[-1], M)]
    got = _steps(hip_device, sizes, x, y, 2, read_hidden=True)
    want = _steps(cpu_oracle, sizes, x, y, 2, read_hidden=True)
    assert_close(np.array(got["loss"]), np.array(want["loss"]), 1e-4, "losses")
    for i, (a, b) in enumerate(zip(got["params"], want["params"])):
        assert_close(a, b, 1e-4, f"param {i}")


def test_input_that_needs_a_gradient_is_not_fused(hip_device, cpu_oracle):
    sizes, M = (48, 64, 7), 24
    rng = np.random.default_rng(6)
    x = rng.random((M, sizes[0]), dtype=np.float32)
    y = np.eye(sizes[-1], dtype=np.float32)[rng.integers(0, sizes[-1], M)]
    got = _steps(hip_device, sizes, x, y, 1, x_grad=True)
    want = _steps(cpu_oracle, sizes, x, y, 1, x_grad=True)
    assert_close(got["dx"], want["dx"], 1e-4, "dx")
    for i, (a, b) in enumerate(zip(got["params"], want["params"])):
        assert_close(a, b, 1e-4, f"param {i}")


def test_captured_mlp_step_replays_the_eager_step(hip_device):
    """bench.py's job: the captured step (hipGraph) of the (784, 256, 10) model at batch 128 equals three eager steps"""
    import bench
    from deepflows_amd import workloads
    from deepflows_amd.graph_step import CapturedTrainStep
    from deepflows_amd.tensor import Graph
    runs = []
    for captured in (False, True):
        model, opt, crit, xt, yt = bench.build_job("mlp", hip_device, 0, 128)
        losses = []
        if captured:
            cap = CapturedTrainStep(model, crit, opt, xt, yt)
            for _ in range(3):
                losses.append(float(cap.replay().numpy()[0]))
            cap.destroy()
        else:
            for _ in range(3):
                loss, _ = workloads.train_step(model, crit, opt, xt, yt)
                losses.append(float(loss.numpy()[0]))
                Graph.free_graph()
        runs.append((losses, [p.data.numpy().copy() for p in model.parameters()]))
        Graph.free_graph_all()
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("sizes,M", [((48, 64, 7), 96), ((160, 128, 16), 256)])
def test_sgd_step_rides_on_the_backward_launch_at_every_hidden_width(hip_device, cpu_oracle, sizes, M):
    """the 64- and 128-wide instantiations of the backward launch with the optimizer's step in it (k_mlp2_bwd<1 / 2, true>)"""
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Graph, Tensor
    rng = np.random.default_rng(M)
    x_np = rng.random((M, sizes[0]), dtype=np.float32)
    y_np = np.eye(sizes[-1], dtype=np.float32)[rng.integers(0, sizes[-1], M)]
    res = {}
    for name, dev in (("hip", hip_device), ("cpu", cpu_oracle)):
        np.random.seed(0)
        model = workloads.MLP(sizes, device=dev)
        model.train()
        opt = optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        crit = nn.CrossEntropyLoss()
        x, y = Tensor(x_np, device=dev), Tensor(y_np, device=dev)
        if name == "hip":
            dev.mod.prof_reset(); dev.mod.prof_enable(True)
        for _ in range(3):
            loss = crit(model(x), y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            Graph.free_graph()
        if name == "hip":
            dev.mod.synchronize(); dev.mod.prof_enable(False)
            fams = {k: v["calls"] for k, v in dev.mod.prof_read().items() if v["calls"]}
            assert fams.get("mlp2_ce_bwd_sgd") == 3 and "sgd_step" not in fams, fams
        res[name] = [p.data.numpy().copy() for p in model.parameters()]
        Graph.free_graph_all()
    for i, (a, b) in enumerate(zip(res["hip"], res["cpu"])):
        assert_close(a, b, 1e-4, f"param {i} after 3 steps")


@pytest.mark.parametrize("opt_kind", ["plain", "momentum_wd", "nesterov"])
def test_sgd_step_rides_on_the_backward_launch(hip_device, cpu_oracle, opt_kind):
    """loss.backward(); opt.step() with nothing in between that reads the gradients: the backward launch is still deferred when
    the optimizer's call arrives and takes the update along (dfhip_sgd_step -> mlp_bwd_take_sgd: optim/sgd.py:16-24 applied to
    every gradient element where it is complete) -- three launches per step, the gradients still written"""
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Graph, Tensor
    sizes, M = (784, 256, 10), 128
    kw = {"plain": {}, "momentum_wd": {"momentum": 0.9, "weight_decay": 5e-4},
          "nesterov": {"momentum": 0.9, "nesterov": True}}[opt_kind]
    rng = np.random.default_rng(12)
    x_np = rng.random((M, sizes[0]), dtype=np.float32)
    y_np = np.eye(sizes[-1], dtype=np.float32)[rng.integers(0, sizes[-1], M)]
    res = {}
    for name, dev in (("hip", hip_device), ("cpu", cpu_oracle)):
        np.random.seed(0)
        model = workloads.MLP(sizes, device=dev)
        model.train()
        opt = optim.SGD(model.parameters(), lr=0.05, **kw)
        crit = nn.CrossEntropyLoss()
        x, y = Tensor(x_np, device=dev), Tensor(y_np, device=dev)
        if name == "hip":
            dev.mod.prof_reset(); dev.mod.prof_enable(True)
        losses = []
        for _ in range(4):
            loss = crit(model(x), y)
            opt.zero_grad()
            loss.backward()
            opt.step()                                   # (the gradients are read only AFTER the step)
            losses.append(float(loss.numpy()[0]))
            grads = [p.grad.numpy().copy() for p in model.parameters()]
            Graph.free_graph()
        if name == "hip":
            dev.mod.synchronize(); dev.mod.prof_enable(False)
            fams = {k: v["calls"] for k, v in dev.mod.prof_read().items() if v["calls"]}
            assert fams.get("mlp2_ce_bwd_sgd") == 4 and "sgd_step" not in fams and "mlp2_ce_bwd" not in fams, fams
        res[name] = (losses, grads, [p.data.numpy().copy() for p in model.parameters()])
        Graph.free_graph_all()
    assert_close(np.array(res["hip"][0]), np.array(res["cpu"][0]), 1e-4, "losses")
    for i, (a, b) in enumerate(zip(res["hip"][1], res["cpu"][1])):
        assert_close(a, b, 1e-4, f"grad {i} of the last step")
    for i, (a, b) in enumerate(zip(res["hip"][2], res["cpu"][2])):
        assert_close(a, b, 1e-4, f"param {i} after 4 steps ({opt_kind})")


def test_step_in_the_launch_has_the_separate_launches_bits(hip_device):
    """the same expressions on the same gradient values: parameters after three steps are bit-identical whether the update
    rode on the backward launch or ran as the multi-tensor launch (the gradients read in between force the latter)"""
    sizes, M = (784, 256, 10), 128
    rng = np.random.default_rng(13)
    x = rng.random((M, sizes[0]), dtype=np.float32)
    y = np.eye(sizes[-1], dtype=np.float32)[rng.integers(0, sizes[-1], M)]
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Graph, Tensor
    outs = []
    for read_between in (False, True):
        np.random.seed(0)
        model = workloads.MLP(sizes, device=hip_device)
        model.train()
        opt = optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        crit = nn.CrossEntropyLoss()
        xt, yt = Tensor(x, device=hip_device), Tensor(y, device=hip_device)
        for _ in range(3):
            loss = crit(model(xt), yt)
            opt.zero_grad()
            loss.backward()
            if read_between:
                model.fc1.weight.grad.numpy()
            opt.step()
            Graph.free_graph()
        outs.append([p.data.numpy().copy() for p in model.parameters()])
        Graph.free_graph_all()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("workload,batch", [("mlp", 128), ("cnn", 64)])
def test_four_steps_per_graph_are_four_replays(hip_device, workload, batch):
    """graph_step.CapturedTrainStep(steps_per_graph=4): eight calls of replay() (two launches of a graph that holds four steps)
    leave the parameters of eight replays of the one-step graph, bit for bit -- Dropout masks drawn in step order (CNN), Adam's
    device table advanced once per step (CNN), the SGD step inside the backward launch (MLP)"""
    import bench
    from deepflows_amd.graph_step import CapturedTrainStep
    from deepflows_amd.tensor import Graph
    runs = []
    for spg in (1, 4):
        np.random.seed(0)
        model, opt, crit, xt, yt = bench.build_job(workload, hip_device, 0, batch)
        np.random.seed(77)
        cap = CapturedTrainStep(model, crit, opt, xt, yt, steps_per_graph=spg)
        for _ in range(8):
            loss = cap.replay()
        last = float(loss.numpy()[0])
        cap.destroy()
        runs.append((last, [p.data.numpy().copy() for p in model.parameters()], opt.t if hasattr(opt, "t") else None))
        Graph.free_graph_all()
    assert runs[0][0] == runs[1][0] and runs[0][2] == runs[1][2], (runs[0][0], runs[1][0], runs[0][2], runs[1][2])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a, b)
