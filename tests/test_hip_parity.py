"""GPU parity tests (pytest -m gpu): the `hip` device — primitives and fused kernels, all
called through the C ABI of libdfhip.so — against the golden vectors generated from the
reference, with the same assertions the oracle passes on CPU."""
import os

import numpy as np
import pytest

import parity_cases as pc
from conftest import assert_close

pytestmark = pytest.mark.gpu


def test_device_is_the_native_extension(hip_device):
    import deepflows_amd._lib as L
    assert hip_device.name == "hip" and hip_device.mod.fused
    assert L._lib is not None, "libdfhip.so must be the thing that runs"
    assert "gfx950" in hip_device.mod.device_name, hip_device.mod.device_name


def test_primitives(hip_device):
    pc.check_primitives(hip_device)


@pytest.mark.parametrize("name", pc.CONV_CASES)
def test_conv2d_overwrite(hip_device, name):
    pc.check_conv2d(hip_device, name, "overwrite")


@pytest.mark.parametrize("name", pc.CONV_CASES)
def test_conv2d_accumulate(hip_device, name):
    pc.check_conv2d(hip_device, name, "accumulate")


def test_batchnorm2d(hip_device):
    pc.check_batchnorm(hip_device)


def test_maxpool2d(hip_device):
    pc.check_maxpool(hip_device)


def test_cross_entropy(hip_device):
    pc.check_cross_entropy(hip_device)


def test_linear_relu_reduce(hip_device):
    pc.check_linear_relu_reduce(hip_device)


@pytest.mark.parametrize("cname", list(pc.OPT_CONFIGS))
def test_optimizers(hip_device, cname):
    pc.check_optimizer(hip_device, cname)


def test_linear_regression_trajectory(hip_device):
    pc.check_linear_regression(hip_device)


def test_mlp_three_sgd_steps(hip_device):
    pc.check_mlp(hip_device)


def test_cnn_two_adam_steps(hip_device):
    pc.check_cnn(hip_device)


@pytest.mark.parametrize("mode", ["f6", "all"])
def test_resnet18_one_adam_step(hip_device, mode):
    pc.check_resnet18(hip_device, mode)


@pytest.mark.parametrize("mode", ["f6", "all"])
def test_resnet18_written_as_the_reference_writes_it(hip_device, mode):
    """the unmodified model text (plain module calls, `out + identity`, two tensor.mean) on the hip device: the modules record
    their calls and launch the fused kernels (nn/lazy.py); same golden as the hand-fused class"""
    pc.check_resnet18(hip_device, mode, plain=True)


@pytest.mark.parametrize("tag", ["vgg", "mbn"])
def test_vgg16_and_mobilenet_one_step(hip_device, tag):
    """VGG16 (test/VGG.py) and MobileNetV1 (test/MobileNet.py) on the fused hip path
    against the reference's own outputs."""
    pc.check_model_step(hip_device, tag)


def test_wide_resnet18_one_step(hip_device):
    """test/ResNet.py's ResNet-18 (64/128/256/512, ReLU after the add) on the fused hip path: against the float64
    yardstick under the measured budget (tests/error_budget.py), and against the reference's own outputs with the
    gradient tolerance that budget gives (never a flat number above 1e-4)."""
    import error_budget as eb
    from conftest import rel_err
    from oracle.numpy_device import NumpyDevice, NumpyDevice64
    exact = eb.with_device(NumpyDevice64, eb.wide_resnet_step)
    fp32 = eb.with_device(NumpyDevice, eb.wide_resnet_step)
    got = eb.wide_resnet_step(hip_device)
    eb.assert_within_budget(got, exact, fp32, "hip wide ResNet-18 step (batch 2)")
    e32 = max(rel_err(fp32[k], exact[k]) for k in ("conv1_wgrad", "fc_wgrad", "grad_abs_sum"))
    pc.check_wide_resnet_step(hip_device, tol_grad=max(1e-4, 4.0 * e32))


@pytest.mark.parametrize("shape", [(8, 32, 16, 16, 32, 3, 1, 1), (6, 32, 16, 16, 64, 3, 2, 1), (4, 64, 8, 8, 64, 3, 1, 1),
                                   (16, 128, 4, 4, 128, 3, 1, 1), (32, 256, 2, 2, 256, 3, 1, 1), (64, 128, 2, 2, 80, 3, 1, 1),
                                   (80, 128, 2, 2, 64, 3, 1, 1), (5, 32, 16, 16, 64, 1, 2, 0),
                                   (8, 64, 8, 8, 128, 1, 2, 0),
                                   (3, 16, 12, 12, 40, 3, 1, 1)])
def test_conv_bn_partials_and_residual(hip_device, shape):
    """conv2d -> BatchNorm2d(train) (+ residual): the statistics come from the partials the
    convolution's epilogue leaves behind; checked against the NumPy restatement of
    nn/modules/batchnorm.py:36-46 on the convolution output (oracle/ref_ops.py)."""
    from oracle import ref_ops
    from deepflows_amd import tensor
    from deepflows_amd.nn import functional as F
    N, C, H, W, O, k, s, p = shape
    rng = np.random.default_rng(11)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((O, C, k, k)) * 0.2).astype(np.float32)
    gamma = rng.standard_normal((1, O, 1, 1)).astype(np.float32)
    beta = rng.standard_normal((1, O, 1, 1)).astype(np.float32)
    y_ref = ref_ops.conv2d_forward(x, w, np.full(O, 0.75, np.float32), s, p)    # non-zero channel means
    res = rng.standard_normal(y_ref.shape).astype(np.float32)
    out_ref, mean_ref, var_ref = ref_ops.batchnorm2d_train(y_ref, gamma, beta, 1e-5)
    dev = hip_device
    mk = lambda a, rg=False: tensor.Tensor(a, device=dev, requires_grad=rg)
    bias = mk(np.full((1, O, 1, 1), 0.75, np.float32))
    xt, wt, gt, bt, rt = mk(x, True), mk(w, True), mk(gamma, True), mk(beta, True), mk(res, True)
    rm, rv = mk(np.zeros((1, O, 1, 1), np.float32)), mk(np.ones((1, O, 1, 1), np.float32))
    y = F.conv2d(xt, wt, padding=p, stride=s, bias=bias)
    eligible = C % 8 == 0 and (H * W) % 4 == 0
    if eligible:
        assert y.bn_partials is not None, "the main forward kernel should leave partials behind"
    for residual in (None, rt):
        out = F.batch_norm2d_train(y, gt, bt, rm, rv, 1e-5, 0.1, residual)
        want = out_ref if residual is None else out_ref + res
        assert_close(out.numpy(), want, 1e-4, "bn(conv)")
        assert_close(out.mean.numpy().reshape(-1), mean_ref.reshape(-1), 1e-4, "mean")
        assert_close(out.var.numpy().reshape(-1), var_ref.reshape(-1), 1e-4, "var")
    # backward through bn(+residual) and the convolution
    g = rng.standard_normal(out_ref.shape).astype(np.float32)
    (out * mk(g)).sum().backward()
    dxh, dgam, dbet = ref_ops.batchnorm2d_backward(g, y_ref, gamma, mean_ref, var_ref, 1e-5)
    assert_close(rt.grad.numpy(), g, 1e-6, "residual grad")
    assert_close(gt.grad.numpy().reshape(-1), dgam.reshape(-1), 1e-4, "dgamma")
    assert_close(bt.grad.numpy().reshape(-1), dbet.reshape(-1), 1e-4, "dbeta")
    assert_close(wt.grad.numpy(), ref_ops.conv2d_wgrad(x, dxh, k, s, p), 1e-4, "dW")


def test_captured_step_matches_eager_steps(hip_device):
    """graph_step.CapturedTrainStep: 1 eager + 3 replayed Adam steps of ResNet-18 leave exactly
    the parameters of 4 eager steps (same kernels in the same order)."""
    import numpy as np
    from deepflows_amd import nn, optim, tensor, workloads
    from deepflows_amd.graph_step import CapturedTrainStep
    from deepflows_amd.tensor import Graph

    def run(captured):
        np.random.seed(5)
        model = workloads.ResNet18(device=hip_device, register_blocks=True)
        opt = optim.Adam(workloads.all_block_parameters(model), lr=1e-3)
        crit = nn.CrossEntropyLoss()
        rng = np.random.default_rng(9)
        x = tensor.Tensor(rng.standard_normal((8, 3, 32, 32)).astype(np.float32), device=hip_device)
        y = tensor.Tensor(np.eye(10, dtype=np.float32)[rng.integers(0, 10, 8)], device=hip_device)
        losses = []
        loss, _ = workloads.train_step(model, crit, opt, x, y)
        losses.append(float(loss.numpy().reshape(-1)[0]))
        Graph.free_graph()
        if captured:
            cs = CapturedTrainStep(model, crit, opt, x, y)
            for _ in range(3):
                losses.append(float(cs.replay().numpy().reshape(-1)[0]))
        else:
            for _ in range(3):
                loss, _ = workloads.train_step(model, crit, opt, x, y)
                losses.append(float(loss.numpy().reshape(-1)[0]))
                Graph.free_graph()
        return losses, [p.numpy().copy() for p in opt.params], opt.t

    l0, p0, t0 = run(False)
    l1, p1, t1 = run(True)
    assert t0 == t1 == 5
    assert l0 == l1, (l0, l1)
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)


def test_scheduler_over_a_captured_sgd_step_matches_eager(hip_device):
    """SGD's scalars are read from a device array inside the captured step, so a scheduler that mutates
    optimizer.lr between replays (optim/scheduler.py:13-61; the MLP script's StepLR) is NOT frozen into the graph:
    captured + scheduler == eager + scheduler bit for bit, and differs from a run whose lr never moves."""
    import numpy as np
    from deepflows_amd import nn, optim, tensor, workloads
    from deepflows_amd.graph_step import CapturedTrainStep
    from deepflows_amd.optim.scheduler import StepLR
    from deepflows_amd.tensor import Graph

    def run(captured, schedule=True, momentum=0.9):
        np.random.seed(11)
        model = workloads.MLP((784, 256, 10), device=hip_device)
        opt = optim.SGD(model.parameters(), lr=0.05, momentum=momentum, weight_decay=5e-4)
        sched = StepLR(opt, step_size=2, gamma=0.5)
        crit = nn.CrossEntropyLoss()
        rng = np.random.default_rng(3)
        x = tensor.Tensor(rng.random((128, 784), dtype=np.float32), device=hip_device)
        y = tensor.Tensor(np.eye(10, dtype=np.float32)[rng.integers(0, 10, 128)], device=hip_device)
        cs = CapturedTrainStep(model, crit, opt, x, y) if captured else None
        lrs = []
        for _ in range(6):
            if captured:
                cs.replay()
            else:
                workloads.train_step(model, crit, opt, x, y)
                Graph.free_graph()
            if schedule:
                sched.step()
            lrs.append(opt.lr)
        return [p.numpy().copy() for p in opt.params], lrs

    for momentum in (0.9, 0.0):
        pe, lre = run(False, momentum=momentum)
        pc, lrc = run(True, momentum=momentum)
        assert lre == lrc and lre[-1] < 0.05 / 4 + 1e-12
        for a, b in zip(pe, pc):
            assert np.array_equal(a, b)
        frozen, _ = run(True, schedule=False, momentum=momentum)
        assert any(not np.array_equal(a, b) for a, b in zip(pc, frozen))


def test_fused_stem_matches_the_three_operators(hip_device):
    """bn_relu_maxpool2_train(x) == max_pool2d(relu(batch_norm2d_train(x)), 2) forward and backward
    (the tie rules of the reference: every tied window maximum gets the gradient — whole windows
    of ReLU zeros — and ReLU passes the gradient where output == input)."""
    from deepflows_amd import nn, tensor
    from deepflows_amd.nn import functional as F
    rng = np.random.default_rng(17)
    for (N, C, H, W) in [(8, 32, 32, 32), (3, 5, 6, 8), (16, 64, 16, 16)]:
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        x[0, 0, :2, :2] = -3.0                      # a window of ReLU zeros: four-way tie
        x[1 % N, 1 % C, 0, 0] = x[1 % N, 1 % C, 0, 1]   # a two-way tie of positive values
        gam = rng.standard_normal((1, C, 1, 1)).astype(np.float32)
        bet = rng.standard_normal((1, C, 1, 1)).astype(np.float32)
        gout = rng.standard_normal((N, C, H // 2, W // 2)).astype(np.float32)
        res = []
        for fused in (False, True):
            bn = nn.BatchNorm2d(C, device=hip_device)
            bn.weight.data = tensor.Tensor(gam, device=hip_device).data
            bn.bias.data = tensor.Tensor(bet, device=hip_device).data
            pool = nn.MaxPool2d(2, 2)
            xt = tensor.Tensor(x, device=hip_device, requires_grad=True)
            if fused:
                assert F.bn_relu_maxpool2_fusable(xt, bn, pool)
                out = F.bn_relu_maxpool2_train(xt, bn)
            else:
                out = pool(F.relu(bn(xt)))
            (out * tensor.Tensor(gout, device=hip_device)).sum().backward()
            res.append((out.numpy(), xt.grad.numpy(), bn.weight.grad.numpy(), bn.bias.grad.numpy(),
                        bn.running_mean.numpy(), bn.running_var.numpy()))
        names = ("y", "dx", "dgamma", "dbeta", "running_mean", "running_var")
        for a, b, nme in zip(res[0], res[1], names):
            assert_close(b, a, 2e-5, f"{nme} {N, C, H, W}")


def test_captured_step_with_dropout_matches_eager_steps(hip_device):
    """The CNN workload has Dropout (mask drawn on the host, nn/modules/dropout.py): in a captured step the mask is
    a graph INPUT, drawn in the reference's RNG order and uploaded in front of every replay -- 1 eager + 3 replayed
    Adam steps leave exactly the losses and parameters of 4 eager steps, and the NumPy generator ends in the same
    state."""
    from deepflows_amd import nn, optim, tensor, workloads
    from deepflows_amd.graph_step import CapturedTrainStep
    from deepflows_amd.tensor import Graph

    def run(captured):
        np.random.seed(3)
        model = workloads.CIFAR10_CNN(10, device=hip_device)
        opt = optim.Adam(model.parameters(), lr=5e-3, weight_decay=5e-4)
        crit = nn.CrossEntropyLoss()
        rng = np.random.default_rng(4)
        x = tensor.Tensor(rng.standard_normal((8, 3, 32, 32)).astype(np.float32), device=hip_device)
        y = tensor.Tensor(np.eye(10, dtype=np.float32)[rng.integers(0, 10, 8)], device=hip_device)
        loss, _ = workloads.train_step(model, crit, opt, x, y)
        losses = [float(loss.numpy().reshape(-1)[0])]
        Graph.free_graph()
        if captured:
            cs = CapturedTrainStep(model, crit, opt, x, y)
            assert len(cs.feeds) >= 1
            for _ in range(3):
                losses.append(float(cs.replay().numpy().reshape(-1)[0]))
        else:
            for _ in range(3):
                loss, _ = workloads.train_step(model, crit, opt, x, y)
                losses.append(float(loss.numpy().reshape(-1)[0]))
                Graph.free_graph()
        return losses, [p.numpy().copy() for p in opt.params], float(np.random.random_sample())

    l0, p0, r0 = run(False)
    l1, p1, r1 = run(True)
    assert l0 == l1, (l0, l1)
    assert r0 == r1, "the captured step consumed the NumPy generator differently"
    assert len(set(l0)) == 4                              # (the masks did change from step to step)
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)


def test_failed_capture_leaves_the_device_usable(hip_device):
    """A forward pass that uploads host data itself cannot be captured (a recorded copy would re-read a recycled
    pinned slot on every replay: dfhip_h2d refuses it); the attempt must raise, end the capture, give the
    capture's blocks back to the pool and leave eager training intact."""
    from deepflows_amd import nn, optim, tensor, workloads
    from deepflows_amd.graph_step import CapturedTrainStep
    from deepflows_amd.tensor import Graph

    class Noisy(nn.Module):
        def __init__(self, dev):
            super().__init__()
            self.fc = nn.Linear(16, 10, device=dev)

        def forward(self, x):
            noise = x.device.empty(x.shape, dtype="float32")
            x.device.from_numpy(np.ones(tuple(x.shape), np.float32), noise._handle)     # host copy in the step
            return self.fc(x * noise)

    np.random.seed(3)
    model = Noisy(hip_device)
    opt = optim.Adam(model.parameters(), lr=1e-3)
    crit = nn.CrossEntropyLoss()
    rng = np.random.default_rng(4)
    x = tensor.Tensor(rng.standard_normal((8, 16)).astype(np.float32), device=hip_device)
    y = tensor.Tensor(np.eye(10, dtype=np.float32)[rng.integers(0, 10, 8)], device=hip_device)
    l0, _ = workloads.train_step(model, crit, opt, x, y)
    Graph.free_graph()
    mod = hip_device.mod
    for attempt in range(3):
        with pytest.raises(RuntimeError):
            CapturedTrainStep(model, crit, opt, x, y)
        Graph.free_graph_all()
        if attempt == 0:
            mallocs = mod.pool_stats()["hipmalloc_calls"]
    assert mod.pool_stats()["hipmalloc_calls"] == mallocs, "a failed capture leaked its blocks"
    opt.device_hyper = False
    losses = []
    for _ in range(3):
        l, _ = workloads.train_step(model, crit, opt, x, y)
        Graph.free_graph()
        losses.append(float(l.numpy()[0]))
    assert all(np.isfinite(v) for v in losses)


def test_adam_notices_a_step_count_the_device_does_not_share(hip_device):
    """The captured Adam update reads its bias corrections from a table the device walks by itself.  When the host's count
    of consumed entries and the device's differ (a replay that was not noted, or noted without running), the next rewrite
    of the table must refuse instead of training on with another step's corrections."""
    from deepflows_amd import nn, optim, tensor, workloads
    from deepflows_amd.graph_step import CapturedTrainStep
    from deepflows_amd.tensor import Graph
    np.random.seed(5)
    model = workloads.MLP((16, 8, 4), device=hip_device)
    opt = optim.Adam(model.parameters(), lr=1e-3)
    crit = nn.CrossEntropyLoss()
    rng = np.random.default_rng(6)
    x = tensor.Tensor(rng.standard_normal((8, 16)).astype(np.float32), device=hip_device)
    y = tensor.Tensor(np.eye(4, dtype=np.float32)[rng.integers(0, 4, 8)], device=hip_device)
    step = CapturedTrainStep(model, crit, opt, x, y)
    for _ in range(3):
        step.replay()
    index, cap = hip_device.mod.adam_hyper_index(opt._hyper)
    assert index == opt._table_used and cap == opt._HYPER_STEPS
    opt.lr = 5e-4                      # a scheduler moved the rate: the table is rewritten in front of the next replay
    step.replay()
    assert hip_device.mod.adam_hyper_index(opt._hyper)[0] == 1
    opt.note_replayed_apply()          # the host counts a step the device never ran
    opt.lr = 2.5e-4
    # (the read-back drains the device: it runs when the table is exhausted, or on every rewrite when asked for)
    from deepflows_amd.optim import adam as adam_module
    adam_module._CHECK_TABLE = True
    try:
        with pytest.raises(RuntimeError, match="host counted"):
            step.replay()
    finally:
        adam_module._CHECK_TABLE = False
    step.destroy()
    Graph.free_graph_all()


@pytest.mark.parametrize("N,C,H,O", [(8, 32, 16, 64), (32, 128, 4, 256)])
def test_paired_batchnorm_matches_two_layers(hip_device, N, C, H, O):
    """bn_a(conv_a(x)) + bn_b(conv_b(x)) as the paired node (one launch each way) against the two separate
    BatchNorm nodes + residual add: outputs, input / parameter gradients and running statistics.  (The second shape ends on
    2 x 2 planes: the kernels with eight channels per workgroup.)"""
    import numpy as np
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor
    F = nn.functional
    dev = hip_device
    rng = np.random.default_rng(3)
    x = rng.standard_normal((N, C, H, H)).astype(np.float32)
    gy = rng.standard_normal((N, O, H // 2, H // 2)).astype(np.float32)
    results = []
    for paired in (True, False):
        np.random.seed(5)
        conv_a = nn.Conv2d(C, O, kernel_size=3, stride=2, padding=1, bias=False, device=dev)
        conv_b = nn.Conv2d(C, O, kernel_size=1, stride=2, padding=0, bias=False, device=dev)
        bn_a, bn_b = nn.BatchNorm2d(O, device=dev), nn.BatchNorm2d(O, device=dev)
        for bn, s in ((bn_a, 1.0), (bn_b, 2.0)):
            bn.train()
            bn.weight.data.device.from_numpy((rng.standard_normal((1, O, 1, 1)) * 0 + s).astype(np.float32), bn.weight.data._handle)
        xt = Tensor(x, device=dev, requires_grad=True)
        za, zb = conv_a(xt), conv_b(xt)
        if paired:
            assert F.batch_norm2d_pair_fusable(za, bn_a, zb, bn_b)
            y = F.batch_norm2d_pair_train(za, bn_a, zb, bn_b)
        else:
            y = bn_a(za, residual=bn_b(zb))
        loss = tensor.sum(y * Tensor(gy, device=dev))
        loss.backward()
        results.append([y.numpy().copy(), xt.grad.numpy().copy(), conv_a.weight.grad.numpy().copy(),
                        conv_b.weight.grad.numpy().copy(), bn_a.weight.grad.numpy().copy(), bn_b.bias.grad.numpy().copy(),
                        bn_a.running_mean.numpy().copy(), bn_b.running_var.numpy().copy()])
        Graph.free_graph_all()
    names = ("y", "dx", "dw_a", "dw_b", "dgamma_a", "dbeta_b", "running_mean_a", "running_var_b")
    for a, b, nm in zip(results[0], results[1], names):
        assert_close(a, b, 2e-5, "paired BatchNorm: " + nm)


@pytest.mark.parametrize("C,HW,dgrad", [(32, 16, "overwrite"), (64, 8, "overwrite"), (128, 4, "overwrite"),
                                         (256, 2, "overwrite"), (64, 8, "accumulate")])
def test_bn_applied_on_the_next_convolutions_loads(hip_device, C, HW, dgrad):
    """conv2(bn1(conv1(x))) of a residual block (no activation in between) with bn1 applied on conv2's operand loads
    (one node, bn1's output never written) against the three separate nodes: output, BatchNorm partials consumed by
    the next BatchNorm, every gradient, running statistics.  Batch 128: the smallest the register-operand kernel takes."""
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor
    F = nn.functional
    dev = hip_device
    mod = dev.mod
    rng = np.random.default_rng(11)
    N = 128
    x = rng.standard_normal((N, C, HW, HW)).astype(np.float32)
    gy = rng.standard_normal((N, C, HW, HW)).astype(np.float32)
    gam = (1.0 + 0.3 * rng.standard_normal((1, C, 1, 1))).astype(np.float32)
    bet = (0.5 * rng.standard_normal((1, C, 1, 1))).astype(np.float32)
    old_mode = mod.conv_dgrad_mode
    mod.conv_dgrad_mode = dgrad
    results = []
    try:
        for fused in (True, False):
            np.random.seed(7)
            conv1 = nn.Conv2d(C, C, kernel_size=3, stride=1, padding=1, bias=False, device=dev)
            conv2 = nn.Conv2d(C, C, kernel_size=3, stride=1, padding=1, bias=False, device=dev)
            bn1, bn2 = nn.BatchNorm2d(C, device=dev), nn.BatchNorm2d(C, device=dev)
            bn1.train(); bn2.train()
            mod.from_numpy(gam, bn1.weight.data._handle)
            mod.from_numpy(bet, bn1.bias.data._handle)
            xt = Tensor(x, device=dev, requires_grad=True)
            z1 = conv1(xt)
            if fused:
                z2 = F.bn_conv2d_train(z1, bn1, conv2)
                assert z2 is not None, "the fused form must take this shape"
                assert getattr(z2, "bn_partials", None) is not None
            else:
                z2 = conv2(bn1(z1))
            y = bn2(z2)                       # consumes z2's partials
            loss = tensor.sum(y * Tensor(gy, device=dev))
            loss.backward()
            results.append([z2.numpy().copy(), y.numpy().copy(), xt.grad.numpy().copy(), conv1.weight.grad.numpy().copy(),
                            conv2.weight.grad.numpy().copy(), bn1.weight.grad.numpy().copy(), bn1.bias.grad.numpy().copy(),
                            bn2.weight.grad.numpy().copy(), bn1.running_mean.numpy().copy(),
                            bn1.running_var.numpy().copy()])
            Graph.free_graph_all()
    finally:
        mod.conv_dgrad_mode = old_mode
    names = ("z2", "y", "dx", "dw1", "dw2", "dgamma1", "dbeta1", "dgamma2", "running_mean1", "running_var1")
    for a, b, nm in zip(results[0], results[1], names):
        assert_close(a, b, 3e-5, f"bn on conv loads C={C}: " + nm)


def test_bn_on_conv_loads_with_a_large_channel_offset(hip_device):
    """the operand path folds bn1 into fma(x, scale, shift) with shift = beta - mean * scale; when |mean| is
    far above the channel's spread the two large terms cancel.  conv1's output gets a channel mean ~500 x its standard
    deviation here (input offset 50, spread 0.1) and conv2(bn1(conv1(x))) is compared with the reference's expression
    evaluated by the oracle in float64 -- allowed: max(1e-4, 2 x what the float32 oracle itself loses)."""
    import error_budget as eb
    from conftest import rel_err
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor
    from oracle.numpy_device import NumpyDevice, NumpyDevice64
    F = nn.functional
    rng = np.random.default_rng(19)
    N, C, HW = 128, 32, 16
    x = (50.0 + 0.1 * rng.standard_normal((N, C, HW, HW))).astype(np.float32)
    gam = (1.0 + 0.3 * rng.standard_normal((1, C, 1, 1))).astype(np.float32)
    bet = (0.5 * rng.standard_normal((1, C, 1, 1))).astype(np.float32)

    def run(dev, fused):
        np.random.seed(7)
        conv1 = nn.Conv2d(C, C, kernel_size=3, stride=1, padding=1, bias=False, device=dev)
        conv2 = nn.Conv2d(C, C, kernel_size=3, stride=1, padding=1, bias=False, device=dev)
        bn1 = nn.BatchNorm2d(C, device=dev)
        bn1.train()
        dev.from_numpy(gam, bn1.weight.data._handle)
        dev.from_numpy(bet, bn1.bias.data._handle)
        z1 = conv1(Tensor(x, device=dev))
        z2 = F.bn_conv2d_train(z1, bn1, conv2) if fused else conv2(bn1(z1))
        assert z2 is not None
        out = {"z1": z1.numpy().astype(np.float64), "z2": z2.numpy().astype(np.float64)}
        Graph.free_graph_all()
        return out
    exact = eb.with_device(NumpyDevice64, lambda d: run(d, False))
    fp32 = eb.with_device(NumpyDevice, lambda d: run(d, False))
    got = run(hip_device, True)
    # only the interior of z1 carries the big offset relative to its spread (the border rows see the zero padding);
    # the ratio this case is about:
    inner = exact["z1"][:, :, 2:-2, 2:-2]
    ratio = np.abs(inner.mean(axis=(0, 2, 3))) / inner.std(axis=(0, 2, 3))
    assert np.median(ratio) > 50, ratio
    e32, e = rel_err(fp32["z2"], exact["z2"]), rel_err(got["z2"], exact["z2"])
    print(f"\nbn on conv loads, |mean| / std of conv1's output {np.median(ratio):.0f} (median): hip {e:.3e}, fp32 oracle {e32:.3e}")
    assert e <= max(1e-4, 2.0 * e32), (e, e32)


def test_wgrad_affine_outside_a_group_writes_the_operand_out(hip_device):
    """dfhip_conv2d_wgrad_affine called stand-alone (no grouped sweep open): the operand x * scale + shift is
    materialised once and the ordinary kernels run; equal to the weight gradient of the explicit operand."""
    mod = hip_device.mod
    rng = np.random.default_rng(5)
    N, C, H, W, O = 4, 8, 6, 6, 8
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    dy = rng.standard_normal((N, O, H, W)).astype(np.float32)
    aff = rng.standard_normal((C, 2)).astype(np.float32)
    def arr(a):
        h = mod.Array(a.size); mod.from_numpy(a, h); return h
    geo = (N, C, H, W, O, 3, 1, 1)
    dw1, dw2 = mod.Array(O * C * 9), mod.Array(O * C * 9)
    mod.conv2d_wgrad_affine(arr(x), arr(aff), arr(dy), dw1, *geo)
    xa = x * aff[None, :, 0, None, None] + aff[None, :, 1, None, None]
    mod.conv2d_wgrad(arr(xa.astype(np.float32)), arr(dy), dw2, *geo)
    a, b = mod.to_numpy(dw1, (dw1.size,), (1,), 0), mod.to_numpy(dw2, (dw2.size,), (1,), 0)
    assert_close(a, b, 1e-5, "wgrad_affine fall-back")


@pytest.mark.parametrize("C,HW,O", [(32, 16, 64), (64, 8, 128), (32, 8, 32), (128, 4, 256), (48, 4, 128), (48, 8, 64)])
def test_shortcut_input_gradient_rides_on_the_stride2_kernel(hip_device, C, HW, O):
    """dfhip_conv2d_dgrad_pair: the reference-exact input gradients of a block's stride-2 3x3 convolution and of its
    1x1 / stride-2 shortcut (both read the block input) as one launch, against the two launches; and the same through
    a residual block's autograd nodes (the shortcut's node then hands nothing back for the input)."""
    mod = hip_device.mod
    rng = np.random.default_rng(13)
    N = 16
    def arr(shape):
        h = mod.Array(int(np.prod(shape))); mod.from_numpy(rng.standard_normal(shape).astype(np.float32), h); return h
    dy, dy2 = arr((N, O, HW // 2, HW // 2)), arr((N, O, HW // 2, HW // 2))
    w, w2, held = arr((O, C, 3, 3)), arr((O, C, 1, 1)), arr((N, C, HW, HW))
    back = lambda h: mod.to_numpy(h, (h.size,), (1,), 0)
    for add in (None, held):
        a, b, t = mod.Array(N * C * HW * HW), mod.Array(N * C * HW * HW), mod.Array(N * C * HW * HW)
        assert mod.conv2d_dgrad_pair(dy, w, dy2, w2, add, a, N, C, HW, HW, O, mode="overwrite")
        if add is None:
            mod.conv2d_dgrad(dy, w, t, N, C, HW, HW, O, 3, 2, 1, mode="overwrite")
        else:
            mod.conv2d_dgrad_add(dy, w, add, t, N, C, HW, HW, O, 3, 2, 1, mode="overwrite")
        mod.conv2d_dgrad_add(dy2, w2, t, b, N, C, HW, HW, O, 1, 2, 0, mode="overwrite")
        assert_close(back(a), back(b), 2e-6, f"dgrad pair C={C} O={O} add={add is not None}")
    # accumulate mode: one launch for the shapes the stride-2 kernel takes (checked against the oracle's complete sums);
    # elsewhere the call reports that it launched nothing
    from oracle import ref_ops as R
    takes = C % 32 == 0 and O in (64, 128, 256) and HW >= 4 and (N * (HW // 2) ** 2) % 128 == 0
    for add in (None, held):
        a = mod.Array(N * C * HW * HW)
        done = mod.conv2d_dgrad_pair(dy, w, dy2, w2, add, a, N, C, HW, HW, O, mode="accumulate")
        assert bool(done) == takes, (C, HW, O)
        if not takes:
            continue
        shape = (N, C, HW, HW)
        want = (R.conv2d_dgrad(back(dy).reshape(N, O, HW // 2, HW // 2), back(w).reshape(O, C, 3, 3), shape, 2, 1, "accumulate")
                + R.conv2d_dgrad(back(dy2).reshape(N, O, HW // 2, HW // 2), back(w2).reshape(O, C, 1, 1), shape, 2, 0, "accumulate"))
        if add is not None:
            want = want + back(held).reshape(shape)
        assert_close(back(a).reshape(shape), want, 1e-4, f"accumulate dgrad pair C={C} O={O} add={add is not None}")


@pytest.mark.parametrize("C,HW,O", [(32, 16, 64), (64, 8, 128), (128, 4, 256)])
def test_stride2_convolution_and_its_shortcut_from_one_launch(hip_device, C, HW, O):
    """dfhip_conv2d_forward_s2_pair (the 1x1 / stride-2 shortcut as a tenth tap of the block's 3x3 / stride-2
    convolution) against the two separate convolutions: both outputs and both sets of BatchNorm partials (through the
    BatchNorm that consumes them), then a whole residual block forward + backward with and without it."""
    from deepflows_amd import nn, tensor, workloads
    from deepflows_amd.tensor import Graph, Tensor
    F = nn.functional
    dev = hip_device
    rng = np.random.default_rng(23)
    N = 256
    x = rng.standard_normal((N, C, HW, HW)).astype(np.float32)
    gy = rng.standard_normal((N, O, HW // 2, HW // 2)).astype(np.float32)
    results = []
    for paired in (True, False):
        F._S2_PAIR = paired
        try:
            np.random.seed(19)
            down = nn.ModuleList([nn.Conv2d(C, O, kernel_size=1, stride=2, bias=False, device=dev), nn.BatchNorm2d(O, device=dev)])
            blk = workloads.ResidualBlock(C, O, stride=2, downsample=down, device=dev)
            blk.train()
            xt = Tensor(x, device=dev, requires_grad=True)
            xin = xt * 1.0
            if paired:
                both = F._Conv2dFused.pair_s2(xin, blk.conv1, down[0])
                assert both is not None, "the paired launch must take this shape"
                z1p, zdp = both[0].numpy().copy(), both[1].numpy().copy()
            else:
                z1p, zdp = blk.conv1(xin).numpy().copy(), down[0](xin).numpy().copy()
            Graph.free_graph_all()
            xt = Tensor(x, device=dev, requires_grad=True)
            xin = xt * 1.0
            y = blk(xin)
            loss = tensor.sum(y * Tensor(gy, device=dev))
            loss.backward()
            results.append([z1p, zdp, y.numpy().copy(), xt.grad.numpy().copy(), blk.conv1.weight.grad.numpy().copy(),
                            down[0].weight.grad.numpy().copy(), blk.bn1.running_var.numpy().copy(),
                            down[1].running_mean.numpy().copy()])
            Graph.free_graph_all()
        finally:
            F._S2_PAIR = True
    names = ("conv1 out", "shortcut out", "block out", "dx", "dw1", "dw_shortcut", "running_var bn1", "running_mean bn_d")
    for a, b, nm in zip(results[0], results[1], names):
        assert_close(a, b, 3e-5, f"s2 pair C={C}: " + nm)


def test_residual_block_with_convolution_shortcut_one_input_gradient_launch(hip_device):
    from deepflows_amd import nn, tensor, workloads
    from deepflows_amd.tensor import Graph, Tensor
    dev = hip_device
    rng = np.random.default_rng(4)
    x = rng.standard_normal((16, 32, 16, 16)).astype(np.float32)
    gy = rng.standard_normal((16, 64, 8, 8)).astype(np.float32)
    results = []
    for linked in (True, False):
        np.random.seed(9)
        down = nn.ModuleList([nn.Conv2d(32, 64, kernel_size=1, stride=2, bias=False, device=dev), nn.BatchNorm2d(64, device=dev)])
        blk = workloads.ResidualBlock(32, 64, stride=2, downsample=down, device=dev)
        blk.train()
        xt = Tensor(x, device=dev, requires_grad=True)
        xin = xt * 1.0                           # a non-leaf input, as inside the network
        y = blk(xin)
        if not linked:
            for node in Graph.node_list:
                if isinstance(node, nn.functional._Conv2dFused):
                    node.shortcut_sibling = None
        loss = tensor.sum(y * Tensor(gy, device=dev))
        loss.backward()
        results.append([y.numpy().copy(), xt.grad.numpy().copy(), blk.conv1.weight.grad.numpy().copy(),
                        down[0].weight.grad.numpy().copy()])
        Graph.free_graph_all()
    for a, b, nm in zip(results[0], results[1], ("y", "dx", "dw1", "dw_shortcut")):
        assert_close(a, b, 2e-6, "residual block, paired input gradient: " + nm)


@pytest.mark.parametrize("k,pad", [(3, 1), (5, 2)])
def test_stem_backward_without_the_full_resolution_gradient(hip_device, k, pad):
    """pool(relu(bn(conv(x)))) with an input that needs no gradient: dfhip_stem_backward (statistics pass + the weight
    gradient that rebuilds d loss / d conv-output while staging it) against the separate backward launches; integer
    data so that ties at the maximum and exact zeros at the ReLU occur."""
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor
    F = nn.functional
    dev = hip_device
    rng = np.random.default_rng(17)
    N = 32
    x = rng.integers(-2, 3, size=(N, 3, 32, 32)).astype(np.float32)
    gy = rng.standard_normal((N, 32, 16, 16)).astype(np.float32)
    results = []
    for fused in (True, False):
        F._STEM_FUSED_BWD = fused
        try:
            np.random.seed(2)
            conv = nn.Conv2d(3, 32, kernel_size=k, stride=1, padding=pad, bias=False, device=dev)
            wq = np.round(conv.weight.numpy() * 8) / 8           # coarse weights: exact ties survive the convolution
            dev.mod.from_numpy(wq.astype(np.float32), conv.weight.data._handle)
            bn = nn.BatchNorm2d(32, device=dev)
            bn.train()
            xt = Tensor(x, device=dev)
            z = conv(xt)
            y = F.bn_relu_maxpool2_train(z, bn)
            loss = tensor.sum(y * Tensor(gy, device=dev))
            loss.backward()
            if fused:
                assert z._dw_from_stem, "the fused backward must take this shape"
            results.append([y.numpy().copy(), conv.weight.grad.numpy().copy(), bn.weight.grad.numpy().copy(),
                            bn.bias.grad.numpy().copy()])
            Graph.free_graph_all()
        finally:
            F._STEM_FUSED_BWD = True
    for a, b, nm in zip(results[0], results[1], ("y", "dw", "dgamma", "dbeta")):
        assert_close(a, b, 2e-6, f"stem backward k={k}: " + nm)


@pytest.mark.parametrize("N,C,H,O,K,bias", [(32, 3, 32, 32, 3, False), (5, 3, 32, 32, 3, True), (7, 2, 16, 20, 3, False),
                                              (3, 1, 8, 32, 3, False), (300, 3, 32, 32, 3, False), (32, 3, 32, 32, 5, True),
                                              (6, 2, 16, 24, 5, False), (260, 3, 32, 32, 5, True), (4, 1, 16, 16, 5, False)])
def test_stem_without_its_convolution_output(hip_device, N, C, H, O, K, bias):
    """pool(relu(bn(conv(x)))) as the one node whose 33 MB convolution output never exists (dfhip_stem.hip: the
    convolution keeps the window extreme that survives bn + relu + max -- the minimum where gamma < 0 --, the backward
    kernels rebuild the convolution from x) against the separate operators with their own backward launches.  Integer
    data and coarse weights: exact ties at the window maximum survive the convolution; gamma has negative and zero
    channels."""
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor
    F = nn.functional
    dev = hip_device
    rng = np.random.default_rng(23)
    x = rng.integers(-2, 3, size=(N, C, H, 32)).astype(np.float32)
    gy = rng.standard_normal((N, O, H // 2, 16)).astype(np.float32)
    gamma = (rng.integers(-4, 9, size=(1, O, 1, 1)) / 4).astype(np.float32)
    gamma.reshape(-1)[:3] = (-1.0, 0.0, 0.75)
    beta = (rng.integers(-4, 5, size=(1, O, 1, 1)) / 8).astype(np.float32)
    pool = nn.MaxPool2d(kernel_size=2, stride=2)
    results = []
    for one_node in (True, False):
        F._STEM_FUSED_BWD = False                    # the comparison path: every operator's own backward launch
        try:
            np.random.seed(2)
            conv = nn.Conv2d(C, O, kernel_size=K, stride=1, padding=K // 2, bias=bias, device=dev)
            wq = np.round(conv.weight.numpy() * 8) / 8
            dev.mod.from_numpy(wq.astype(np.float32), conv.weight.data._handle)
            if bias:
                dev.mod.from_numpy((np.arange(O, dtype=np.float32).reshape(1, O, 1, 1) % 5 - 2) / 4, conv.bias.data._handle)
            bn = nn.BatchNorm2d(O, device=dev)
            dev.mod.from_numpy(gamma, bn.weight.data._handle)
            dev.mod.from_numpy(beta, bn.bias.data._handle)
            bn.train()
            xt = Tensor(x, device=dev)
            if one_node:
                y = F.stem_conv_bn_relu_pool_train(xt, conv, bn, pool)
                assert y is not None, "the one-node stem must take this shape"
            else:
                y = F.bn_relu_maxpool2_train(conv(xt), bn)
            loss = tensor.sum(y * Tensor(gy, device=dev))
            loss.backward()
            results.append([y.numpy().copy(), bn.running_mean.numpy().copy(), bn.running_var.numpy().copy(),
                            conv.weight.grad.numpy().copy(), bn.weight.grad.numpy().copy(), bn.bias.grad.numpy().copy(),
                            conv.bias.grad.numpy().copy() if bias else np.zeros(1, np.float32)])
            Graph.free_graph_all()
        finally:
            F._STEM_FUSED_BWD = True
    names = ("y", "running_mean", "running_var", "dw", "dgamma", "dbeta")
    for a, b, nm in zip(results[0], results[1], names):
        assert_close(a, b, 3e-6 if nm in ("y", "running_mean", "running_var") else 2e-5, f"stem without z ({N},{C},{H},{O},k{K}): " + nm)
    if bias:
        # the bias gradient is the sum of d loss / d z, which BatchNorm makes zero up to rounding: both paths return
        # rounding noise far below the weight gradient's scale
        scale = float(np.abs(results[1][3]).max())
        assert float(np.abs(results[0][6]).max()) <= 1e-4 * scale and float(np.abs(results[1][6]).max()) <= 1e-4 * scale
    # the pooled values are relu(bn(.)) of an element of the window: where the two paths' statistics agree to the bit
    # the outputs must too -- at least the zero pattern (ReLU) is identical
    assert np.array_equal(results[0][0] == 0.0, results[1][0] == 0.0)


@pytest.mark.parametrize("m,k,p", [(128, 784, 256), (128, 256, 10), (37, 100, 20), (256, 2048, 10)])
def test_linear_relu_as_one_node(hip_device, m, k, p):
    """relu(linear(x, W, b)) as one forward and one backward launch (the ReLU mask on the operand loads) against the two
    nodes; integer data, so pre-activations that are exactly 0 occur and must pass the gradient (tensor.py:872-877)."""
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor
    F = nn.functional
    rng = np.random.default_rng(3)
    x = rng.integers(-2, 3, size=(m, k)).astype(np.float32)
    w = (rng.integers(-2, 3, size=(k, p)) / 4).astype(np.float32)
    b = (rng.integers(-2, 3, size=(1, p)) / 2).astype(np.float32)
    g = rng.standard_normal((m, p)).astype(np.float32)
    results = []
    for fused in (True, False):
        xt, wt, bt = (Tensor(a, device=hip_device, requires_grad=True) for a in (x, w, b))
        y = F.linear_relu(xt, wt, bt) if fused else F.relu(F.linear(xt, wt, bt))
        if fused:
            assert type(y).__name__ == "_LinearReluFused", "the one-node form must take this shape"
        tensor.sum(y * Tensor(g, device=hip_device)).backward()
        results.append([y.numpy().copy(), xt.grad.numpy().copy(), wt.grad.numpy().copy(), bt.grad.numpy().copy()])
        Graph.free_graph_all()
    z = x.astype(np.float64) @ w + b
    assert (z == 0).sum() > 0, "the case must contain exact zeros"
    for a, c, nm in zip(results[0], results[1], ("y", "dx", "dw", "db")):
        assert_close(a, c, 2e-6, f"linear_relu {m}x{k}x{p}: " + nm)
    assert np.array_equal(results[0][0], np.maximum(z, 0).astype(np.float32))          # exact in this integer case


def test_classifier_head_as_one_node(hip_device):
    """fc(mean(mean(x, 2), 2)) + cross-entropy: the pooled-linear node and the loss gradient left behind by the loss
    kernel, against the separate nodes (mean_hw, Linear, cross-entropy with its own backward launch)."""
    from deepflows_amd import nn, tensor
    from deepflows_amd.tensor import Graph, Tensor
    F = nn.functional
    dev = hip_device
    rng = np.random.default_rng(21)
    for (N, C, H, W, P) in ((256, 256, 2, 2, 10), (6, 20, 3, 3, 7), (5, 300, 1, 1, 16), (300, 70, 2, 2, 3)):
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        y = np.eye(P, dtype=np.float32)[rng.integers(0, P, N)] * 0.95 + 0.05 / P
        results = []
        for fused in (True, False):
            np.random.seed(3)
            fc = nn.Linear(C, P, device=dev)
            xt = Tensor(x, device=dev, requires_grad=True)
            if fused:
                out = F.mean_hw_linear(xt, fc)
                assert type(out).__name__ == "_PoolLinearFused"
                loss = F.cross_entropy(out, Tensor(y, device=dev))
                assert loss.unit_grad is not None
            else:
                out = fc(F.mean_hw(xt))
                loss = F.cross_entropy(out, Tensor(y, device=dev))
                loss.unit_grad = None                      # the loss gradient from its own backward launch
            loss.backward()
            results.append([out.numpy().copy(), loss.numpy().copy(), xt.grad.numpy().copy(), fc.weight.grad.numpy().copy(),
                            fc.bias.grad.numpy().copy()])
            Graph.free_graph_all()
        for a, b, nm in zip(results[0], results[1], ("logits", "loss", "dx", "dW", "db")):
            assert_close(a, b, 2e-6, f"head {N, C, H, W, P}: " + nm)


def test_loss_gradient_left_by_the_forward_kernel(hip_device):
    """dfhip_softmax_ce_forward_grad: dlogits for an upstream gradient of 1 equals dfhip_softmax_ce_backward with a
    unit seed, including rows whose maximum is tied (the reference routes -sum(g) to every tied maximum)."""
    mod = hip_device.mod
    rng = np.random.default_rng(9)
    rows, cols = 37, 10
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    x[3, 2] = x[3, 7] = x[3].max() + 1.0                    # a tie at the maximum
    t = (np.eye(cols, dtype=np.float32)[rng.integers(0, cols, rows)] * 0.9 + 0.01).astype(np.float32)
    def arr(a):
        h = mod.Array(a.size); mod.from_numpy(a, h); return h
    xl, tl = arr(x), arr(t)
    loss1, loss2, d1, d2 = mod.Array(1), mod.Array(1), mod.Array(rows * cols), mod.Array(rows * cols)
    assert mod.softmax_ce_forward_grad(xl, tl, loss1, d1, rows, cols)
    mod.softmax_ce_forward(xl, tl, loss2, rows, cols)
    mod.softmax_ce_backward(xl, tl, arr(np.ones(1, np.float32)), d2, rows, cols)
    back = lambda h: mod.to_numpy(h, (h.size,), (1,), 0)
    assert np.array_equal(back(loss1), back(loss2))
    assert_close(back(d1), back(d2), 1e-6, "dlogits")


def test_avg_pool_activation_modules_and_nll(hip_device):
    """SURVEY §8(f) row 3 on the hip device (26 hand-written primitives, no fused kernels): AvgPool2d incl. the
    zero-padding divisor and the last-writer-wins backward, LeakyReLU / Tanh / Softmax / LogSoftmax, nll_loss."""
    pc.check_avg_pool(hip_device)
    pc.check_activation_modules(hip_device)
    pc.check_nll_loss(hip_device)


def test_frozen_parameter_step(hip_device):
    """test/ResNet18_parameter_freezing_test.py:100-216 ('partial'): one Adam step with the stem and layer1-2 frozen."""
    pc.check_frozen_parameter_step(hip_device)
