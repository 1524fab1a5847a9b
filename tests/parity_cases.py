"""Parity checks against the golden fixtures, parametrised by device so the SAME assertions
run (a) on the NumPy oracle device on CPU (validating the oracle + front-end restatement
against the reference's own outputs) and (b) on the `hip` device on an MI355X (the parity
tests proper, through the C ABI)."""
import numpy as np

import deepflows_amd as df
from deepflows_amd import nn, optim, tensor, workloads
from deepflows_amd.nn import functional as F
from deepflows_amd.tensor import Graph, Tensor

from conftest import assert_close, load_golden

TOL = 1e-4          # BASELINE: fp32 within 1e-4 relative


def T(a, dev, rg=False):
    return Tensor(np.asarray(a, dtype=np.float32), device=dev, requires_grad=rg)


def backward_with(y, g, dev):
    loss = tensor.sum(y * T(g, dev))
    loss.backward()


# ---------------------------------------------------------------------------------------------
def check_primitives(dev):
    g = load_golden("g1_primitives.npz")
    mod = dev.mod

    def arr(np_a):
        h = mod.Array(np_a.size)
        mod.from_numpy(np_a, h)
        return h

    def back(h, shape=None):
        n = h.size
        out = mod.to_numpy(h, (n,), (1,), 0)
        return out if shape is None else out.reshape(shape)

    a, b = arr(g["a"]), arr(g["b"])
    for name in ("add", "mul", "div", "maximum"):
        out = mod.Array(24)
        getattr(mod, "ewise_" + name)(a, b, out)
        assert_close(back(out), g["ewise_" + name], 1e-6, "ewise_" + name)
    for name in ("eq", "ge"):                       # comparison masks are bit-exact
        out = mod.Array(24)
        getattr(mod, "ewise_" + name)(a, b, out)
        assert np.array_equal(back(out), g["ewise_" + name]), "ewise_" + name
    for name in ("add", "mul", "div", "power", "maximum"):
        out = mod.Array(24)
        getattr(mod, "scalar_" + name)(a, float(g["scalar_" + name + "_s"]), out)
        assert_close(back(out), g["scalar_" + name], 1e-6, "scalar_" + name)
    for name in ("eq", "ge"):
        out = mod.Array(24)
        getattr(mod, "scalar_" + name)(a, float(g["scalar_" + name + "_s"]), out)
        assert np.array_equal(back(out), g["scalar_" + name]), "scalar_" + name
    out = mod.Array(24)
    mod.scalar_power(arr(g["pos"]), 0.5, out)
    assert_close(back(out), g["scalar_power_half"], 1e-6, "scalar_power 0.5")
    out = mod.Array(6)
    mod.ewise_log(arr(g["log_in"]), out)
    got, want = back(out), g["ewise_log"]
    assert np.isneginf(got[0]) and np.isnan(got[1]), "log(0) = -inf, log(<0) = nan (NumPy semantics)"
    assert_close(got[2:], want[2:], 1e-6, "ewise_log")
    out = mod.Array(24); mod.ewise_exp(a, out); assert_close(back(out), g["ewise_exp"], 1e-6, "ewise_exp")
    out = mod.Array(24); mod.ewise_tanh(a, out); assert_close(back(out), g["ewise_tanh"], 1e-6, "ewise_tanh")
    for k in ("perm", "slice", "neg", "bcast"):
        v = g["compact_" + k + "_view"]
        nd = (len(v) - 1) // 2
        shp, st, off = tuple(v[:nd]), tuple(v[nd:2 * nd]), int(v[-1])
        out = mod.Array(int(np.prod(shp)))
        mod.compact(a, out, shp, st, off)
        assert np.array_equal(back(out, shp), g["compact_" + k]), "compact " + k      # copies: bit-exact
        assert np.array_equal(mod.to_numpy(a, shp, st, off), g["compact_" + k]), "to_numpy " + k
    dst = mod.Array(24)
    mod.fill(dst, 0.0)
    mod.ewise_setitem(arr(np.arange(1, 5, dtype=np.float32)), dst, (2, 2), (4, 1), 5)
    mod.scalar_setitem(3, 9.0, dst, (3,), (6,), 2)
    assert np.array_equal(back(dst), g["setitem"]), "scatter setitem placement"
    for rs in (1, 4, 10, 256):
        src = arr(g[f"reduce_in_{rs}"])
        o1 = mod.Array(3); mod.reduce_sum(src, o1, rs)
        o2 = mod.Array(3); mod.reduce_max(src, o2, rs)
        assert_close(back(o1), g[f"reduce_sum_{rs}"], 1e-5, f"reduce_sum {rs}")
        assert np.array_equal(back(o2), g[f"reduce_max_{rs}"]), f"reduce_max {rs}"
    out = mod.Array(15)
    mod.matmul(arr(g["mm_a"]), arr(g["mm_b"]), out, 5, 7, 3)
    assert_close(back(out, (5, 3)), g["mm_out"], 1e-5, "matmul")
    h = mod.Array(5); mod.fill(h, 3.14); assert np.array_equal(back(h), g["kat_fill"])
    o = mod.Array(5)
    mod.ewise_add(arr(np.array([1, 2, 3, 4, 5], np.float32)), arr(np.array([10, 20, 30, 40, 50], np.float32)), o)
    assert np.array_equal(back(o), g["kat_ewise_add"])
    o = mod.Array(3); mod.scalar_add(arr(np.array([1, 2, 3], np.float32)), 5.0, o)
    assert np.array_equal(back(o), g["kat_scalar_add"])


CONV_CASES = ["k3s1p1", "k3s2p1", "k1s2p0", "k5s1p2", "k3s1p1_odd", "k3s1p0", "k2s2p0"]


def check_conv2d(dev, name, mode="overwrite"):
    g = load_golden("g2_conv2d.npz")
    N, C, H, W, O, k, s, p = (int(v) for v in g[name + "_geo"])
    has_b = (name + "_b") in g
    if hasattr(dev.mod, "conv_dgrad_mode"):
        dev.mod.conv_dgrad_mode = mode
    xt, wt = T(g[name + "_x"], dev, True), T(g[name + "_w"], dev, True)
    bt = T(g[name + "_b"], dev, True) if has_b else None
    y = F.conv2d(xt, wt, p, s, bias=bt)
    assert_close(y.numpy(), g[name + "_y"], TOL, name + " forward")
    backward_with(y, g[name + "_gy"], dev)
    assert_close(wt.grad.numpy(), g[name + "_dw"], TOL, name + " wgrad")
    assert_close(xt.grad.numpy(), g[name + "_dx_" + mode], TOL, name + " dgrad " + mode)
    if has_b:
        assert_close(bt.grad.numpy(), g[name + "_db"], TOL, name + " bias grad")
    Graph.free_graph_all()


def check_batchnorm(dev):
    g = load_golden("g3_batchnorm2d.npz")
    C = g["gamma"].shape[1]
    bn = nn.BatchNorm2d(C, device=dev)
    bn.weight.data.device.from_numpy(g["gamma"], bn.weight.data._handle)
    bn.bias.data.device.from_numpy(g["beta"], bn.bias.data._handle)
    bn.train()
    xt = T(g["x1"], dev, True)
    y = bn(xt)
    assert_close(y.numpy(), g["y1"], TOL, "bn forward")
    assert_close(bn.running_mean.numpy(), g["rm1"], TOL, "running_mean after 1")
    assert_close(bn.running_var.numpy(), g["rv1"], TOL, "running_var after 1")
    backward_with(y, g["gy"], dev)
    assert_close(xt.grad.numpy(), g["dx"], TOL, "bn dx")
    assert_close(bn.weight.grad.numpy(), g["dgamma"], TOL, "bn dgamma")
    assert_close(bn.bias.grad.numpy(), g["dbeta"], TOL, "bn dbeta")
    y2 = bn(T(g["x2"], dev))
    assert_close(y2.numpy(), g["y2"], TOL, "bn forward 2")
    assert_close(bn.running_mean.numpy(), g["rm2"], TOL, "running_mean after 2")
    assert_close(bn.running_var.numpy(), g["rv2"], TOL, "running_var after 2")
    bn.eval()
    assert_close(bn(T(g["x1"], dev)).numpy(), g["y_eval"], TOL, "bn eval")
    bn.train()
    Graph.free_graph_all()


def check_maxpool(dev):
    g = load_golden("g4_maxpool2d.npz")
    xt = T(g["x"], dev, True)
    y = F.max_pool2d(xt, 2, 2, 0)
    assert np.array_equal(y.numpy(), g["y"]), "maxpool forward is a selection: bit-exact"
    backward_with(y, g["gy"], dev)
    assert_close(xt.grad.numpy(), g["dx"], 1e-6, "maxpool backward (ties get the gradient)")
    Graph.free_graph_all()


def check_cross_entropy(dev):
    g = load_golden("g5_cross_entropy.npz")
    for tag in ("onehot", "smooth"):
        lt = T(g["logits"], dev, True)
        loss = F.cross_entropy(lt, T(g[tag], dev))
        assert loss.shape == (1,)
        assert_close(loss.numpy(), g["loss_" + tag], TOL, "CE loss " + tag)
        loss.backward()
        assert_close(lt.grad.numpy(), g["dlogits_" + tag], TOL, "CE dlogits " + tag)
        Graph.free_graph_all()
    assert_close(F.softmax(T(g["logits"], dev), 1).numpy(), g["softmax"], TOL, "softmax")


def check_linear_relu_reduce(dev):
    g = load_golden("g6_linear_relu_reduce.npz")
    xt, wt, bt = T(g["lin_x"], dev, True), T(g["lin_w"], dev, True), T(g["lin_b"], dev, True)
    pre = F.linear(xt, wt, bt)
    assert_close(pre.numpy(), g["lin_pre"], TOL, "linear")
    y = F.relu(pre)
    assert_close(y.numpy(), g["lin_y"], TOL, "relu(linear)")
    backward_with(y, g["lin_gy"], dev)
    assert_close(xt.grad.numpy(), g["lin_dx"], TOL, "linear dx")
    assert_close(wt.grad.numpy(), g["lin_dw"], TOL, "linear dw")
    assert_close(bt.grad.numpy(), g["lin_db"], TOL, "linear db")
    Graph.free_graph_all()
    xt = T(g["relu_x"], dev, True)
    y = F.relu(xt)
    assert np.array_equal(y.numpy(), g["relu_y"])
    backward_with(y, g["relu_g"], dev)
    assert np.array_equal(xt.grad.numpy(), g["relu_dx"]), "relu passes the gradient at x == 0"
    Graph.free_graph_all()
    xt = T(g["mean_x"], dev, True)
    m1 = tensor.mean(xt, axis=2)
    m2 = tensor.mean(m1, axis=2)
    assert_close(m1.numpy(), g["mean_1"], 1e-6, "mean(axis=2) divides by the total count")
    assert_close(m2.numpy(), g["mean_2"], 1e-6, "second mean")
    backward_with(m2, g["mean_g"], dev)
    assert_close(xt.grad.numpy(), g["mean_dx"], 1e-5, "mean backward")
    Graph.free_graph_all()
    xt = T(g["sum_x"], dev, True)
    s = tensor.sum(tensor.sum(xt, axes=0, keepdims=True), axes=2, keepdims=True)
    assert_close(s.numpy(), g["sum_y"], 1e-5, "sum keepdims")
    backward_with(s, g["sum_g"], dev)
    assert_close(xt.grad.numpy(), g["sum_dx"], 1e-6, "sum backward")
    Graph.free_graph_all()
    pt = T(g["mse_p"], dev, True)
    l = F.mse_loss(pt, T(g["mse_t"], dev))
    assert_close(l.numpy(), g["mse_loss"], 1e-5, "mse")
    l.backward()
    assert_close(pt.grad.numpy(), g["mse_dp"], 1e-5, "mse grad")
    Graph.free_graph_all()


OPT_CONFIGS = {
    "sgd_plain": lambda ps: optim.SGD(ps, lr=0.05),
    "sgd_mom_wd": lambda ps: optim.SGD(ps, lr=0.05, momentum=0.9, weight_decay=5e-4),
    "sgd_nesterov": lambda ps: optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True),
    "adam_wd": lambda ps: optim.Adam(ps, lr=1e-3, weight_decay=5e-4),
    "adam_plain": lambda ps: optim.Adam(ps, lr=5e-3),
}


def check_optimizer(dev, cname):
    g = load_golden("g7_optimizers.npz")
    ps = [T(g[f"p0_{i}"], dev, True) for i in range(3)]
    opt = OPT_CONFIGS[cname](ps)
    for t in range(3):
        for i, p in enumerate(ps):
            p.grad = df.backend_api.Btensor(g[f"g{t}_{i}"], device=dev)
        opt.step()
        for i, p in enumerate(ps):
            assert_close(p.data.numpy(), g[f"{cname}_step{t + 1}_{i}"], 1e-5, f"{cname} step {t + 1} param {i}")
    Graph.free_graph_all()


# ---- end to end ---------------------------------------------------------------------------------
def check_linear_regression(dev):
    g = load_golden("g8_end_to_end.npz")
    np.random.seed(42)
    X_np = np.random.rand(100, 1) * 1
    y_np = 3 * X_np + 5 + np.random.randn(100, 1)
    model = workloads.LinearRegressionModel(device=dev)
    crit = nn.MSELoss()
    opt = optim.Adam(model.parameters(), lr=0.05, weight_decay=0.001)
    Xt, yt = Tensor(X_np, device=dev), Tensor(y_np, device=dev)
    losses = []
    for _ in range(5):
        loss = crit(model(Xt), yt)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.numpy().item())
    assert_close(np.array(losses), g["linreg_losses"], TOL, "LinearRegression loss trajectory")
    assert_close(model.linear.weight.data.numpy(), g["linreg_w"], TOL, "w")
    assert_close(model.linear.bias.data.numpy(), g["linreg_b"], TOL, "b")
    Graph.free_graph_all()


def check_mlp(dev):
    g = load_golden("g8_end_to_end.npz")
    np.random.seed(0)
    model = workloads.MLP((784, 256, 10), device=dev)
    opt = optim.SGD(model.parameters(), lr=0.05)
    crit = nn.CrossEntropyLoss()
    losses = []
    for step in range(3):
        out = model(Tensor(g["mlp_x"], device=dev))
        loss = crit(out, Tensor(g["mlp_y"], device=dev))
        opt.zero_grad(); loss.backward()
        if step == 0:
            assert_close(out.numpy(), g["mlp_logits0"], TOL, "MLP logits")
            gn = np.array([np.linalg.norm(p.grad.numpy()) for p in model.parameters()])
            assert_close(gn, g["mlp_gnorms0"], TOL, "MLP grad norms")
        opt.step()
        losses.append(loss.numpy().item())
        Graph.free_graph()
    assert_close(np.array(losses), g["mlp_losses"], TOL, "MLP losses")
    assert_close(model.fc2.bias.data.numpy(), g["mlp_final_fc2_b"], TOL, "MLP fc2 bias after 3 steps")
    Graph.free_graph_all()


def check_cnn(dev):
    g = load_golden("g8_end_to_end.npz")
    np.random.seed(0)
    model = workloads.CIFAR10_CNN(10, device=dev)
    model.train()
    assert len(list(model.parameters())) == int(g["cnn_n_params"][0])
    opt = optim.Adam(model.parameters(), lr=0.005, weight_decay=5e-4)
    crit = nn.CrossEntropyLoss()
    losses = []
    for step in range(2):
        out = model(Tensor(g["cnn_x"], device=dev))
        loss = crit(out, Tensor(g["cnn_y"], device=dev))
        opt.zero_grad(); loss.backward()
        if step == 0:
            assert_close(out.numpy(), g["cnn_logits0"], TOL, "CNN logits")
            gn = np.array([np.linalg.norm(p.grad.numpy()) for p in model.parameters()])
            assert_close(gn, g["cnn_gnorms0"], TOL, "CNN grad norms")
            assert_close(model.conv1.weight.grad.numpy(), g["cnn_conv1_wgrad0"], TOL, "CNN conv1 wgrad")
        opt.step()
        losses.append(loss.numpy().item())
        Graph.free_graph()
    assert_close(np.array(losses), g["cnn_losses"], TOL, "CNN losses (2 Adam steps)")
    assert_close(model.bn1.running_mean.numpy(), g["cnn_bn1_rm"], TOL, "CNN bn1 running mean")
    assert_close(model.bn1.running_var.numpy(), g["cnn_bn1_rv"], TOL, "CNN bn1 running var")
    Graph.free_graph_all()


def check_resnet18(dev, mode, plain=False):
    """plain: the model class whose forward() is the reference script's text (workloads.ResNet18Plain)"""
    g = load_golden("g8_end_to_end.npz")
    np.random.seed(0)
    model = (workloads.ResNet18Plain if plain else workloads.ResNet18)(10, device=dev, register_blocks=(mode == "all"))
    model.train()
    params_all = workloads.all_block_parameters(model)
    opt_params = list(model.parameters())
    n, numel = (int(v) for v in g[f"r18_{mode}_n_opt_params"])
    assert len(opt_params) == n and sum(int(np.prod(p.shape)) for p in opt_params) == numel
    assert len(params_all) == 62
    init = np.array([float(np.abs(p.data.numpy()).sum()) for p in params_all])
    assert_close(init, g[f"r18_{mode}_init_checksum"], 1e-6, "seeded init reproduces the reference's weights")
    opt = optim.Adam(opt_params, lr=1e-3, weight_decay=5e-4)
    out = model(Tensor(g["r18_x"], device=dev))
    loss = nn.CrossEntropyLoss()(out, Tensor(g["r18_y"], device=dev))
    opt.zero_grad(); loss.backward()
    assert_close(loss.numpy(), g[f"r18_{mode}_loss"], TOL, "R18 loss")
    assert_close(out.numpy(), g[f"r18_{mode}_logits"], TOL, "R18 logits")
    assert_close(model.fc.weight.grad.numpy(), g[f"r18_{mode}_fc_wgrad"], TOL, "R18 fc wgrad")
    assert_close(model.fc.bias.grad.numpy(), g[f"r18_{mode}_fc_bgrad"], TOL, "R18 fc bgrad")
    assert_close(model.conv1.weight.grad.numpy(), g[f"r18_{mode}_conv1_wgrad"], TOL, "R18 conv1 wgrad (end of a 20-conv chain)")
    gabs = np.array([float(np.abs(p.grad.numpy()).sum()) for p in params_all])
    assert_close(gabs, g[f"r18_{mode}_grad_abs_sum"], TOL, "R18 |grad| checksums of all 62 tensors")
    opt.step()
    after = np.array([float(np.abs(p.data.numpy()).sum()) for p in params_all])
    assert_close(after, g[f"r18_{mode}_param_abs_sum_after"], TOL, "R18 parameters after the Adam step")
    assert_close(model.bn1.running_mean.numpy(), g[f"r18_{mode}_bn1_rm"], TOL, "R18 bn1 running mean")
    assert_close(model.bn1.running_var.numpy(), g[f"r18_{mode}_bn1_rv"], TOL, "R18 bn1 running var")
    Graph.free_graph_all()


def check_model_step(dev, tag, tol_grad=TOL):
    """One seeded training step of VGG16 (tag 'vgg', Adam) or MobileNetV1 x0.25 (tag 'mbn', SGD with
    momentum) against outputs of the reference's own classes (oracle/gen_models_golden.py)."""
    g = load_golden("g10_models.npz")
    np.random.seed(0)
    if tag == "vgg":
        model = workloads.VGG16(10, device=dev, img_size=(32, 32))
        first = model.conv1_1
    else:
        model = workloads.MobileNetV1(num_classes=10, width_multiplier=0.25, device=dev, img_size=(32, 32))
        first = model.conv1
    model.train()
    params = list(model.parameters())
    n, numel = (int(v) for v in g[f"{tag}_n_params"])
    assert len(params) == n and sum(int(np.prod(p.shape)) for p in params) == numel
    init = np.array([float(np.abs(p.data.numpy()).sum()) for p in params])
    assert_close(init, g[f"{tag}_init_abs_sum"], 1e-6, f"{tag}: seeded init reproduces the reference's weights")
    if tag == "vgg":
        opt = optim.Adam(params, lr=1e-3, weight_decay=5e-4)
    else:
        opt = optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=0.0005)
    out = model(Tensor(g["x"], device=dev))
    loss = nn.CrossEntropyLoss()(out, Tensor(g["y"], device=dev))
    opt.zero_grad(); loss.backward()
    assert_close(out.numpy(), g[f"{tag}_logits"], TOL, f"{tag} logits")
    assert_close(loss.numpy(), g[f"{tag}_loss"], TOL, f"{tag} loss")
    gabs = np.array([float(np.abs(p.grad.numpy()).sum()) for p in params])
    assert_close(gabs, g[f"{tag}_grad_abs_sum"], tol_grad, f"{tag} |grad| checksums")
    assert_close(first.weight.grad.numpy(), g[f"{tag}_conv1_wgrad"], tol_grad, f"{tag} first conv wgrad")
    opt.step()
    after = np.array([float(np.abs(p.data.numpy()).sum()) for p in params])
    assert_close(after, g[f"{tag}_param_abs_sum_after"], 1e-4, f"{tag} parameters after the step")
    Graph.free_graph_all()


def check_wide_resnet_step(dev, tol_grad=TOL):
    """One SGD step of the repository's other ResNet-18 (test/ResNet.py:24-154: 64/128/256/512 channels, no stem
    pool, ReLU after the residual add; blocks in Python lists, so the optimizer holds 5 of the 62 tensors) at batch 2
    against the reference's own outputs (oracle/gen_models_golden.py, `wide_*`)."""
    g = load_golden("g10_models.npz")
    np.random.seed(0)
    model = workloads.ResNet18Wide(10, device=dev, register_blocks=False)
    model.train()
    every = workloads.all_block_parameters(model)
    n, numel, n_registered = (int(v) for v in g["wide_n_params"])
    params = list(model.parameters())
    assert len(every) == n and sum(int(np.prod(p.shape)) for p in every) == numel and len(params) == n_registered
    init = np.array([float(np.abs(p.data.numpy()).sum()) for p in every])
    assert_close(init, g["wide_init_abs_sum"], 1e-6, "wide: seeded init reproduces the reference's weights")
    opt = optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=0.0005)
    out = model(Tensor(g["x"], device=dev))
    loss = nn.CrossEntropyLoss()(out, Tensor(g["y"], device=dev))
    opt.zero_grad(); loss.backward()
    assert_close(out.numpy(), g["wide_logits"], TOL, "wide logits")
    assert_close(loss.numpy(), g["wide_loss"], TOL, "wide loss")
    gabs = np.array([float(np.abs(p.grad.numpy().astype(np.float64)).sum()) for p in every])
    assert_close(gabs, g["wide_grad_abs_sum"], tol_grad, "wide |grad| checksums (all 62 tensors)")
    assert_close(model.conv1.weight.grad.numpy(), g["wide_conv1_wgrad"], tol_grad, "wide first conv wgrad")
    assert_close(model.fc.weight.grad.numpy(), g["wide_fc_wgrad"], tol_grad, "wide fc wgrad")
    opt.step()
    after = np.array([float(np.abs(p.data.numpy().astype(np.float64)).sum()) for p in every])
    assert_close(after, g["wide_param_abs_sum_after"], 1e-4, "wide parameters after the step")
    assert_close(model.bn1.running_var.numpy().reshape(-1), g["wide_running_var_bn1"].reshape(-1), TOL, "wide running var")
    Graph.free_graph_all()


# ---------------------------------------------------------------------------------------------
# the remaining nn surface against outputs of the reference (g11_nn_surface.npz,
# oracle/gen_f3_golden.py)
def check_avg_pool(dev):
    g = load_golden("g11_nn_surface.npz")
    # the reference's own avg_pool2d raises (functional.py:402 calls a Tensor method that does not exist); the
    # fixture holds its steps executed with its primitives and the documented window mean (see the generator)
    assert "has no attribute 'mean'" in str(g["ap_reference_error"])
    for name in ("ap_k2s2", "ap_k3s1p1", "ap_k3s2p1"):
        k, s, p = (int(v) for v in g[name + "_geo"])
        for via_module in (False, True):
            xt = T(g[name + "_x"], dev, True)
            y = nn.AvgPool2d(k, s, p)(xt) if via_module else F.avg_pool2d(xt, k, s, p)
            assert_close(y.numpy(), g[name + "_y"], 1e-5, name + " forward")
            backward_with(y, g[name + "_gy"], dev)
            # overlapping windows: the last window that covers a pixel wins (functional.py:285-294)
            assert_close(xt.grad.numpy(), g[name + "_dx"], 1e-5, name + " backward")
            Graph.free_graph_all()


def check_activation_modules(dev):
    g = load_golden("g11_nn_surface.npz")
    for name, m in (("leaky", nn.LeakyReLU(0.1)), ("tanh", nn.Tanh()), ("softmax", nn.Softmax(dim=1)),
                    ("logsoftmax", nn.LogSoftmax(dim=1))):
        xt = T(g["act_x"], dev, True)
        y = m(xt)
        assert_close(y.numpy(), g[name + "_y"], 1e-5, name + " forward")
        backward_with(y, g["act_gy"], dev)
        assert_close(xt.grad.numpy(), g[name + "_dx"], 2e-5, name + " backward")
        Graph.free_graph_all()


def check_nll_loss(dev):
    g = load_golden("g11_nn_surface.npz")
    for red in ("mean", "sum"):
        for via_module in (False, True):
            lt = T(g["nll_logp"], dev, True)
            tg = T(g["nll_target"], dev)
            loss = nn.NLLLoss(reduction=red)(lt, tg) if via_module else F.nll_loss(lt, tg, red)
            loss.backward()
            assert_close(loss.numpy(), g[f"nll_{red}"], 1e-5, f"nll {red}")
            assert_close(lt.grad.numpy(), g[f"nll_{red}_dlogp"], 1e-6, f"nll {red} backward")
            Graph.free_graph_all()


def frozen_parameter_sets(model):
    """The parameter lists of test/ResNet18_parameter_freezing_test.py:100-216, strategy 'partial', on the
    reference's ResNet (blocks in Python lists, shortcut layers in a Python list inside the block: neither is
    reached by `parameters()`, so `block.parameters()` is the block's own two conv + BatchNorm pairs)."""
    def block_params(layer):
        out = []
        for blk in layer:
            out += list(blk.parameters())
        return out
    every = list(model.conv1.parameters()) + list(model.bn1.parameters())
    for layer in (model.layer1, model.layer2, model.layer3, model.layer4):
        every += block_params(layer)
    every += list(model.fc.parameters())
    frozen = list(model.conv1.parameters()) + list(model.bn1.parameters()) + block_params(model.layer1) + \
        block_params(model.layer2)
    return every, frozen


def check_frozen_parameter_step(dev, tol_grad=TOL):
    """One Adam step of ResNet-18 with conv1 / bn1 / layer1 / layer2 frozen (requires_grad = False) and the
    optimizer over filter(requires_grad): frozen parameters receive no gradient and do not move, the others
    match the reference's step."""
    g = load_golden("g11_nn_surface.npz")
    np.random.seed(0)
    model = workloads.ResNet18(10, device=dev, register_blocks=False)
    model.train()
    every, frozen = frozen_parameter_sets(model)
    assert [len(every), len(frozen)] == [int(v) for v in g["frz_n_params"][:2]]
    assert len(list(model.parameters())) == int(g["frz_n_params"][2])
    for p in every:
        p.requires_grad = True
    for p in frozen:
        p.requires_grad = False
    opt = optim.Adam(list(filter(lambda p: p.requires_grad, every)), lr=1e-3, weight_decay=5e-4)
    before = [p.numpy().copy() for p in every]
    assert_close(np.array([float(np.abs(b).sum()) for b in before]), g["frz_abs_sum_before"], 1e-6, "seeded init")
    out = model(T(g["frz_x"], dev))
    loss = nn.CrossEntropyLoss()(out, T(g["frz_y"], dev))
    opt.zero_grad()
    loss.backward()
    assert_close(loss.numpy(), g["frz_loss"], TOL, "loss")
    assert_close(out.numpy(), g["frz_logits"], TOL, "logits")
    has_grad = np.array([p.grad is not None for p in every])
    assert np.array_equal(has_grad, g["frz_has_grad"]), "which parameters received a gradient"
    gsum = np.array([float(np.abs(p.grad.numpy()).sum()) if p.grad is not None else 0.0 for p in every])
    assert_close(gsum, g["frz_grad_abs_sum"], tol_grad, "gradient |.| sums of the trainable parameters")
    opt.step()
    frozen_ids = {id(p) for p in frozen}
    for p, b in zip(every, before):
        if id(p) in frozen_ids:
            assert np.array_equal(p.numpy(), b), "a frozen parameter moved"
    assert_close(np.array([float(np.abs(p.numpy()).sum()) for p in every]), g["frz_abs_sum_after"], 1e-5, "after Adam")
    assert_close(model.fc.weight.numpy(), g["frz_fc_w_after"], TOL, "fc.weight after Adam")
    Graph.free_graph_all()
