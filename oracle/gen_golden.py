"""TEST INFRASTRUCTURE — golden-vector generator.  Runs ONLY in the build container, where the
reference is mounted at /root/reference: imports the reference's own Python (DeepFlows/),
executes its device primitives, layers, optimizers and training steps on seeded inputs and
writes inputs + expected outputs to tests/golden/*.npz.  The reference never travels to the
GPU box; the fixtures do.

Oracle tiers:
  O1  the unmodified reference on its `cpu` device — everything that never hits a strided
      __setitem__ (matmul, ewise/scalar, reduce, compact, Linear, ReLU, softmax/CE, BatchNorm2d,
      SGD/Adam, mean);
  O2  the reference's Python stack with its NumpyBackend's two setitem methods replaced by the
      scatter semantics of its own CUDA device (ndarray_backend_cuda.cu:178-221) — pad, im2col,
      Conv2d, MaxPool2d; reproduces the last-writer-wins input gradient (functional.py:285-294);
  O3  torch (CPU, this container only) for the mathematically correct ("accumulate") conv
      input gradient.

Model classes are lifted from the reference's test scripts at run time with `ast` (the scripts
themselves execute data loading at import and hard-code Windows paths); no reference source is
written into this repository.

    python oracle/gen_golden.py            # regenerates tests/golden/
"""
import ast
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import DeepFlows  # noqa: F401
    from DeepFlows.backend import backend_tensor as bt

    orig_cpu_numpy = bt.cpu_numpy

    def _lin(shape, strides, offset):
        idx = np.indices(tuple(int(s) for s in shape), dtype=np.int64)
        return int(offset) + sum(idx[d] * int(strides[d]) for d in range(len(shape)))

    def cpu_numpy_o2():
        dev = orig_cpu_numpy()
        cls = type(dev.mod)

        def ewise_setitem(self, other_handle, view_handle, shape, strides, offset):
            view_handle[_lin(shape, strides, offset).reshape(-1)] = other_handle.reshape(-1)

        def scalar_setitem(self, size, scalar, view_handle, shape, strides, offset):
            view_handle[_lin(shape, strides, offset).reshape(-1)] = np.float32(scalar)

        cls.ewise_setitem = ewise_setitem
        cls.scalar_setitem = scalar_setitem
        return dev

    return bt, orig_cpu_numpy, cpu_numpy_o2


def lift_classes(script, names, replace=()):
    src = open(os.path.join(REF, "test", script), encoding="utf-8").read()
    tree = ast.parse(src)
    chunks = []
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names:
            chunks.append(ast.get_source_segment(src, node))
    code = "\n\n".join(chunks)
    for a, b in replace:
        code = code.replace(a, b)
    return code


def main():
    os.makedirs(OUT, exist_ok=True)
    bt, cpu_o1, cpu_o2 = import_reference()
    import DeepFlows
    from DeepFlows import tensor as rt
    from DeepFlows import nn as rnn
    from DeepFlows.nn import functional as RF
    from DeepFlows import optim as roptim
    from DeepFlows.tensor import Tensor as RT
    from DeepFlows import backend_api as rapi

    def use(tier):
        bt.cpu_numpy = cpu_o2 if tier == "O2" else cpu_o1
        return rapi.Device("cpu")

    def T(a, dev, rg=False):
        return RT(np.asarray(a, dtype=np.float32), device=dev, requires_grad=rg)

    def backward_with(y, g, dev):
        """inject upstream gradient g: loss = sum(y * g)"""
        loss = rt.sum(y * T(g, dev))
        loss.backward()

    # ------------------------------------------------------------------ G1 primitives (O1 + O2 setitem)
    rng = np.random.default_rng(101)
    g1 = {}
    dev = use("O2")
    mod = dev.mod
    a = rng.standard_normal(24).astype(np.float32)
    b = rng.standard_normal(24).astype(np.float32)
    b[3] = a[3]; b[7] = a[7]
    g1["a"], g1["b"] = a, b
    for name in ("add", "mul", "div", "maximum", "eq", "ge"):
        out = mod.Array(24); getattr(mod, "ewise_" + name)(a, b, out); g1["ewise_" + name] = out.copy()
    for name, s in (("add", 1.5), ("mul", -2.0), ("div", 3.0), ("power", 2.0), ("maximum", 0.1),
                    ("eq", float(a[3])), ("ge", 0.0)):
        out = mod.Array(24); getattr(mod, "scalar_" + name)(a, s, out)
        g1["scalar_" + name] = out.copy(); g1["scalar_" + name + "_s"] = np.float32(s)
    pos = np.abs(a) + 0.5
    out = mod.Array(24); mod.scalar_power(pos, 0.5, out); g1["pos"] = pos; g1["scalar_power_half"] = out.copy()
    lg_in = np.array([0.0, -1.0, 1.0, 2.5, 1e-30, 7.0], dtype=np.float32)
    out = mod.Array(6)
    with np.errstate(all="ignore"):
        mod.ewise_log(lg_in, out)
    g1["log_in"], g1["ewise_log"] = lg_in, out.copy()
    out = mod.Array(24); mod.ewise_exp(a, out); g1["ewise_exp"] = out.copy()
    out = mod.Array(24); mod.ewise_tanh(a, out); g1["ewise_tanh"] = out.copy()
    # compact of permuted / sliced / offset / negative-stride / stride-0 views of a (2,3,4) array
    views = {"perm": ((4, 2, 3), (1, 12, 4), 0), "slice": ((2, 2, 2), (12, 4, 2), 5),
             "neg": ((2, 3, 4), (12, 4, -1), 3), "bcast": ((3, 2, 4), (0, 12, 1), 4)}
    for k, (shp, st, off) in views.items():
        out = mod.Array(int(np.prod(shp))); mod.compact(a, out, shp, st, off)
        g1["compact_" + k] = out.copy().reshape(shp)
        g1["compact_" + k + "_view"] = np.array(list(shp) + list(st) + [off], dtype=np.int64)
    # scatter setitem (O2 semantics)
    dst = mod.Array(24); dst[:] = 0
    mod.ewise_setitem(np.arange(1, 5, dtype=np.float32), dst, (2, 2), (4, 1), 5)
    mod.scalar_setitem(3, 9.0, dst, (3,), (6,), 2)
    g1["setitem"] = dst.copy()
    for rs in (1, 4, 10, 256):
        src = rng.standard_normal(3 * rs).astype(np.float32)
        o1 = mod.Array(3); mod.reduce_sum(src, o1, rs)
        o2 = mod.Array(3); mod.reduce_max(src, o2, rs)
        g1[f"reduce_in_{rs}"], g1[f"reduce_sum_{rs}"], g1[f"reduce_max_{rs}"] = src, o1.copy(), o2.copy()
    A = rng.standard_normal((5, 7)).astype(np.float32)
    B = rng.standard_normal((7, 3)).astype(np.float32)
    out = mod.Array(15); mod.matmul(A.reshape(-1), B.reshape(-1), out, 5, 7, 3)
    g1["mm_a"], g1["mm_b"], g1["mm_out"] = A, B, out.copy().reshape(5, 3)
    # the three known-answer tests of test/test_cuda.py:36-99
    h = mod.Array(5); mod.fill(h, 3.14); g1["kat_fill"] = h.copy()
    o = mod.Array(5); mod.ewise_add(np.array([1, 2, 3, 4, 5], np.float32), np.array([10, 20, 30, 40, 50], np.float32), o)
    g1["kat_ewise_add"] = o.copy()
    o = mod.Array(3); mod.scalar_add(np.array([1, 2, 3], np.float32), 5.0, o); g1["kat_scalar_add"] = o.copy()
    np.savez_compressed(os.path.join(OUT, "g1_primitives.npz"), **g1)

    # ------------------------------------------------------------------ G2 conv2d (O2 + O3)
    import torch
    import torch.nn.functional as TF
    g2 = {}
    cases = [("k3s1p1", 2, 4, 8, 8, 8, 3, 1, 1, True), ("k3s2p1", 2, 4, 8, 8, 8, 3, 2, 1, False),
             ("k1s2p0", 3, 8, 8, 8, 4, 1, 2, 0, False), ("k5s1p2", 2, 3, 8, 8, 4, 5, 1, 2, True),
             ("k3s1p1_odd", 2, 3, 7, 7, 5, 3, 1, 1, False), ("k3s1p0", 2, 2, 6, 6, 3, 3, 1, 0, False),
             ("k2s2p0", 2, 3, 8, 8, 4, 2, 2, 0, True)]
    dev = use("O2")
    rng = np.random.default_rng(202)
    for name, N, C, H, W, O, k, s, p, has_b in cases:
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        w = (rng.standard_normal((O, C, k, k)) * 0.3).astype(np.float32)
        b = rng.standard_normal((1, O, 1, 1)).astype(np.float32) if has_b else None
        xt, wt = T(x, dev, True), T(w, dev, True)
        y = RF.conv2d(xt, wt, p, s)
        bt_ = None
        if has_b:
            bt_ = T(b, dev, True)
            y = y + bt_
        yv = y.numpy()
        gy = rng.standard_normal(yv.shape).astype(np.float32)
        backward_with(y, gy, dev)
        g2[name + "_geo"] = np.array([N, C, H, W, O, k, s, p], dtype=np.int64)
        g2[name + "_x"], g2[name + "_w"], g2[name + "_y"], g2[name + "_gy"] = x, w, yv, gy
        g2[name + "_dw"], g2[name + "_dx_overwrite"] = wt.grad.numpy(), xt.grad.numpy()
        if has_b:
            g2[name + "_b"], g2[name + "_db"] = b, bt_.grad.numpy()
        # O3: torch reference for forward cross-check and the accumulate input gradient
        xt3 = torch.tensor(x, requires_grad=True); wt3 = torch.tensor(w, requires_grad=True)
        y3 = TF.conv2d(xt3, wt3, torch.tensor(b.reshape(-1)) if has_b else None, stride=s, padding=p)
        y3.backward(torch.tensor(gy))
        assert np.abs(y3.detach().numpy() - yv).max() < 1e-4, name
        assert np.abs(wt3.grad.numpy() - g2[name + "_dw"]).max() < 1e-4, name
        g2[name + "_dx_accumulate"] = xt3.grad.numpy()
        rt.Graph.free_graph_all()
    np.savez_compressed(os.path.join(OUT, "g2_conv2d.npz"), **g2)

    # ------------------------------------------------------------------ G3 BatchNorm2d (O1)
    g3 = {}
    dev = use("O1")
    rng = np.random.default_rng(303)
    N, C, H, W = 4, 3, 5, 5
    x1 = (rng.standard_normal((N, C, H, W)) * 2 + 0.5).astype(np.float32)
    x2 = (rng.standard_normal((N, C, H, W)) - 1.0).astype(np.float32)
    gam = rng.standard_normal((1, C, 1, 1)).astype(np.float32)
    bet = rng.standard_normal((1, C, 1, 1)).astype(np.float32)
    gy = rng.standard_normal((N, C, H, W)).astype(np.float32)
    bn = rnn.BatchNorm2d(C, device="cpu")
    bn.weight.data.device.from_numpy(gam, bn.weight.data._handle)
    bn.bias.data.device.from_numpy(bet, bn.bias.data._handle)
    bn.train()
    xt = T(x1, dev, True)
    y = bn(xt)
    g3["y1"] = y.numpy()
    g3["rm1"], g3["rv1"] = bn.running_mean.numpy(), bn.running_var.numpy()
    backward_with(y, gy, dev)
    g3["dx"], g3["dgamma"], g3["dbeta"] = xt.grad.numpy(), bn.weight.grad.numpy(), bn.bias.grad.numpy()
    y2 = bn(T(x2, dev, False))
    g3["y2"] = y2.numpy()
    g3["rm2"], g3["rv2"] = bn.running_mean.numpy(), bn.running_var.numpy()
    bn.eval()
    g3["y_eval"] = bn(T(x1, dev)).numpy()
    bn.train()
    g3.update(x1=x1, x2=x2, gamma=gam, beta=bet, gy=gy)
    rt.Graph.free_graph_all()
    np.savez_compressed(os.path.join(OUT, "g3_batchnorm2d.npz"), **g3)

    # ------------------------------------------------------------------ G4 MaxPool2d (O2), G5 CE / softmax, G6 linear/relu/mean/sum (O1)
    g4 = {}
    dev = use("O2")
    rng = np.random.default_rng(404)
    x = rng.integers(-3, 4, size=(2, 3, 6, 6)).astype(np.float32)      # small integers => ties
    x[0, 0, 0:2, 0:2] = 2.0
    gy = rng.standard_normal((2, 3, 3, 3)).astype(np.float32)
    xt = T(x, dev, True)
    y = RF.max_pool2d(xt, 2, 2, 0)
    g4["y"] = y.numpy()
    backward_with(y, gy, dev)
    g4.update(x=x, gy=gy, dx=xt.grad.numpy())
    rt.Graph.free_graph_all()
    np.savez_compressed(os.path.join(OUT, "g4_maxpool2d.npz"), **g4)

    g5 = {}
    dev = use("O1")
    rng = np.random.default_rng(505)
    logits = (rng.standard_normal((8, 10)) * 3).astype(np.float32)
    logits[2, 4] = logits[2].max() + 0.0; logits[2, 1] = logits[2, 4]          # a tie at the max
    labels = rng.integers(0, 10, 8)
    onehot = np.eye(10, dtype=np.float32)[labels]
    smooth = (onehot * 0.95 + 0.05 / 10).astype(np.float32)
    for tag, tgt in (("onehot", onehot), ("smooth", smooth)):
        lt = T(logits, dev, True)
        loss = RF.cross_entropy(lt, T(tgt, dev))
        g5["loss_" + tag] = loss.numpy()
        loss.backward()
        g5["dlogits_" + tag] = lt.grad.numpy()
        rt.Graph.free_graph_all()
    g5["softmax"] = RF.softmax(T(logits, dev), 1).numpy()
    g5.update(logits=logits, onehot=onehot, smooth=smooth)
    np.savez_compressed(os.path.join(OUT, "g5_cross_entropy.npz"), **g5)

    g6 = {}
    rng = np.random.default_rng(606)
    x = rng.standard_normal((6, 5)).astype(np.float32)
    w = rng.standard_normal((5, 4)).astype(np.float32)
    b = rng.standard_normal((1, 4)).astype(np.float32)
    xt, wt, bt2 = T(x, dev, True), T(w, dev, True), T(b, dev, True)
    pre = RF.linear(xt, wt, bt2)
    pre_np = pre.numpy()
    y = RF.relu(pre)
    gy = rng.standard_normal((6, 4)).astype(np.float32)
    backward_with(y, gy, dev)
    g6.update(lin_x=x, lin_w=w, lin_b=b, lin_pre=pre_np, lin_y=y.numpy(), lin_gy=gy,
              lin_dx=xt.grad.numpy(), lin_dw=wt.grad.numpy(), lin_db=bt2.grad.numpy())
    rt.Graph.free_graph_all()
    # relu with exact zeros in the input: x == 0 passes the gradient
    xz = np.array([[-1.0, 0.0, 2.0, 0.0], [0.0, 3.0, -0.5, 0.0]], dtype=np.float32)
    gz = np.array([[1.0, 2.0, 3.0, 4.0], [5.0, 6.0, 7.0, 8.0]], dtype=np.float32)
    xt = T(xz, dev, True)
    y = RF.relu(xt)
    backward_with(y, gz, dev)
    g6.update(relu_x=xz, relu_g=gz, relu_y=y.numpy(), relu_dx=xt.grad.numpy())
    rt.Graph.free_graph_all()
    # mean(axis) forward / backward (divides by the TOTAL count), as used for the ResNet GAP
    xm = np.arange(48, dtype=np.float32).reshape(2, 3, 4, 2)
    xt = T(xm, dev, True)
    m1 = rt.mean(xt, axis=2)
    m2 = rt.mean(m1, axis=2)
    gm = rng.standard_normal((2, 3)).astype(np.float32)
    backward_with(m2, gm, dev)
    g6.update(mean_x=xm, mean_1=m1.numpy(), mean_2=m2.numpy(), mean_g=gm, mean_dx=xt.grad.numpy())
    rt.Graph.free_graph_all()
    xs = rng.standard_normal((3, 4, 5)).astype(np.float32)
    xt = T(xs, dev, True)
    s = rt.sum(rt.sum(xt, axes=0, keepdims=True), axes=2, keepdims=True)
    gs = rng.standard_normal((1, 4, 1)).astype(np.float32)
    backward_with(s, gs, dev)
    g6.update(sum_x=xs, sum_y=s.numpy(), sum_g=gs, sum_dx=xt.grad.numpy())
    rt.Graph.free_graph_all()
    # mse_loss (LinearRegression config)
    p_ = rng.standard_normal((10, 1)).astype(np.float32); t_ = rng.standard_normal((10, 1)).astype(np.float32)
    pt = T(p_, dev, True)
    l = RF.mse_loss(pt, T(t_, dev))
    l.backward()
    g6.update(mse_p=p_, mse_t=t_, mse_loss=l.numpy(), mse_dp=pt.grad.numpy())
    rt.Graph.free_graph_all()
    np.savez_compressed(os.path.join(OUT, "g6_linear_relu_reduce.npz"), **g6)

    # ------------------------------------------------------------------ G7 optimizers (O1)
    g7 = {}
    rng = np.random.default_rng(707)
    shapes = [(4, 3), (1, 3), (2, 2, 3, 3)]
    p0 = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    grads = [[rng.standard_normal(s).astype(np.float32) for s in shapes] for _ in range(3)]
    for i, a_ in enumerate(p0):
        g7[f"p0_{i}"] = a_
        for t_ in range(3):
            g7[f"g{t_}_{i}"] = grads[t_][i]
    configs = {
        "sgd_plain": lambda ps: roptim.SGD(ps, lr=0.05),
        "sgd_mom_wd": lambda ps: roptim.SGD(ps, lr=0.05, momentum=0.9, weight_decay=5e-4),
        "sgd_nesterov": lambda ps: roptim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True),
        "adam_wd": lambda ps: roptim.Adam(ps, lr=1e-3, weight_decay=5e-4),
        "adam_plain": lambda ps: roptim.Adam(ps, lr=5e-3),
    }
    for cname, make in configs.items():
        ps = [T(a_, dev, True) for a_ in p0]
        opt = make(ps)
        for t_ in range(3):
            for p_, g_ in zip(ps, grads[t_]):
                p_.grad = rapi.Btensor(g_, device=dev)
            opt.step()
            for i, p_ in enumerate(ps):
                g7[f"{cname}_step{t_ + 1}_{i}"] = p_.data.numpy()
        rt.Graph.free_graph_all()
    np.savez_compressed(os.path.join(OUT, "g7_optimizers.npz"), **g7)

    # ------------------------------------------------------------------ G8 end-to-end seeded steps
    g8 = {}
    # LinearRegression (O1): test/LinearRegression.py:19-52, first 5 iterations
    dev = use("O1")
    bt.cpu_numpy = cpu_o1
    np.random.seed(42)
    X_np = np.random.rand(100, 1) * 1
    y_np = 3 * X_np + 5 + np.random.randn(100, 1)
    ns = {"nn": rnn, "__name__": "lifted"}
    exec(lift_classes("LinearRegression.py", {"LinearRegressionModel"}), ns)
    model = ns["LinearRegressionModel"]()
    crit = rnn.MSELoss()
    opt = roptim.Adam(model.parameters(), lr=0.05, weight_decay=0.001)
    Xt, yt = RT(X_np), RT(y_np)
    losses = []
    for _ in range(5):
        out = model(Xt)
        loss = crit(out, yt)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.numpy().item())
    g8["linreg_losses"] = np.array(losses, dtype=np.float64)
    g8["linreg_w"] = model.linear.weight.data.numpy(); g8["linreg_b"] = model.linear.bias.data.numpy()
    rt.Graph.free_graph_all()

    # MLP 784-256-10, bs 128, SGD lr .05, 3 steps (O1)
    class RefMLP(rnn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = rnn.Linear(784, 256, device="cpu")
            self.fc2 = rnn.Linear(256, 10, device="cpu")

        def forward(self, x):
            return self.fc2(RF.relu(self.fc1(x)))

    np.random.seed(0)
    model = RefMLP()
    drng = np.random.default_rng(1234)
    xb = drng.random((128, 784), dtype=np.float32)
    yb = np.eye(10, dtype=np.float32)[drng.integers(0, 10, 128)]
    opt = roptim.SGD(model.parameters(), lr=0.05)
    crit = rnn.CrossEntropyLoss()
    losses = []
    for step in range(3):
        out = model(RT(xb, device=dev))
        loss = crit(out, RT(yb, device=dev))
        opt.zero_grad(); loss.backward()
        if step == 0:
            g8["mlp_logits0"] = out.numpy()
            g8["mlp_gnorms0"] = np.array([np.linalg.norm(p.grad.numpy()) for p in model.parameters()])
        opt.step()
        losses.append(loss.numpy().item())
        rt.Graph.free_graph()
    g8["mlp_losses"] = np.array(losses)
    g8["mlp_final_fc2_b"] = model.fc2.bias.data.numpy()
    g8["mlp_x"], g8["mlp_y"] = xb, yb
    rt.Graph.free_graph_all()

    # CNN-CIFAR10 bs 4, 2 Adam steps (O2): test/CNN_CIFAR10_cuda.py:61-108, Adam lr 5e-3 wd 5e-4
    dev = use("O2")
    ns = {"nn": rnn, "__name__": "lifted"}
    exec(lift_classes("CNN_CIFAR10_cuda.py", {"CIFAR10_CNN"}, replace=[("'cuda'", "'cpu'")]), ns)
    np.random.seed(0)
    model = ns["CIFAR10_CNN"](10)
    model.train()
    drng = np.random.default_rng(1234)
    xb = np.clip(drng.standard_normal((4, 3, 32, 32)).astype(np.float32), -1, 1)
    yb = np.eye(10, dtype=np.float32)[drng.integers(0, 10, 4)]
    opt = roptim.Adam(model.parameters(), lr=0.005, weight_decay=5e-4)
    crit = rnn.CrossEntropyLoss()
    losses = []
    for step in range(2):
        out = model(RT(xb, device=dev))
        loss = crit(out, RT(yb, device=dev))
        opt.zero_grad(); loss.backward()
        if step == 0:
            g8["cnn_logits0"] = out.numpy()
            g8["cnn_gnorms0"] = np.array([np.linalg.norm(p.grad.numpy()) for p in model.parameters()])
            g8["cnn_conv1_wgrad0"] = model.conv1.weight.grad.numpy()
        opt.step()
        losses.append(loss.numpy().item())
        rt.Graph.free_graph()
    g8["cnn_losses"] = np.array(losses)
    g8["cnn_bn1_rm"], g8["cnn_bn1_rv"] = model.bn1.running_mean.numpy(), model.bn1.running_var.numpy()
    g8["cnn_x"], g8["cnn_y"] = xb, yb
    g8["cnn_n_params"] = np.array([len(list(model.parameters()))])
    rt.Graph.free_graph_all()

    # ResNet-18 CIFAR variant bs 2, 1 Adam step (O2): test/ResNet_CIFAR10_cuda.py:20-100,166,181-183
    code = lift_classes("ResNet_CIFAR10_cuda.py", {"ResidualBlock", "ResNet", "ResNet18"})
    glb = {"nn": rnn, "tensor": rt, "Tensor": RT, "__name__": "lifted"}
    exec(code, glb)

    def all_params(model):
        seen, out = set(), []

        def visit(m):
            for p in m._parameters.values():
                if p is not None and id(p) not in seen:
                    seen.add(id(p)); out.append(p)
            for c in m._modules.values():
                if c is not None:
                    visit(c)
            for v in m.__dict__.values():
                if isinstance(v, (list, tuple)):
                    for it in v:
                        if isinstance(it, rnn.Module):
                            visit(it)
        visit(model)
        return sorted(out, key=lambda p: p.unique_id)     # construction order

    drng = np.random.default_rng(1234)
    xb = np.clip(drng.standard_normal((2, 3, 32, 32)).astype(np.float32), -1, 1)
    yb = (np.eye(10, dtype=np.float32)[drng.integers(0, 10, 2)] * 0.95 + 0.005).astype(np.float32)
    for mode in ("f6", "all"):
        np.random.seed(0)
        model = glb["ResNet18"](10, device="cpu")
        model.train()
        params_all = all_params(model)
        reg = list(model.parameters())
        opt_params = reg if mode == "f6" else params_all
        g8[f"r18_{mode}_n_opt_params"] = np.array([len(opt_params), sum(int(np.prod(p.shape)) for p in opt_params)])
        g8[f"r18_{mode}_init_checksum"] = np.array([float(np.abs(p.data.numpy()).sum()) for p in params_all])
        opt = roptim.Adam(opt_params, lr=1e-3, weight_decay=5e-4)
        out = model(RT(xb, device=dev))
        loss = rnn.CrossEntropyLoss()(out, RT(yb, device=dev))
        opt.zero_grad(); loss.backward()
        g8[f"r18_{mode}_loss"] = loss.numpy()
        g8[f"r18_{mode}_logits"] = out.numpy()
        g8[f"r18_{mode}_grad_abs_sum"] = np.array([float(np.abs(p.grad.numpy()).sum()) for p in params_all])
        g8[f"r18_{mode}_grad_sum"] = np.array([float(p.grad.numpy().sum()) for p in params_all])
        g8[f"r18_{mode}_conv1_wgrad"] = model.conv1.weight.grad.numpy()
        g8[f"r18_{mode}_fc_wgrad"] = model.fc.weight.grad.numpy()
        g8[f"r18_{mode}_fc_bgrad"] = model.fc.bias.grad.numpy()
        opt.step()
        g8[f"r18_{mode}_param_abs_sum_after"] = np.array([float(np.abs(p.data.numpy()).sum()) for p in params_all])
        g8[f"r18_{mode}_bn1_rm"] = model.bn1.running_mean.numpy()
        g8[f"r18_{mode}_bn1_rv"] = model.bn1.running_var.numpy()
        rt.Graph.free_graph_all()
    g8["r18_x"], g8["r18_y"] = xb, yb
    np.savez_compressed(os.path.join(OUT, "g8_end_to_end.npz"), **g8)

    # ------------------------------------------------------------------ appendix B: primitive-call histograms
    import collections
    hist = {}

    def counting(factory):
        def f():
            devc = factory()
            cls = type(devc.mod)
            for name in list(vars(cls)):
                if name.startswith("_"):
                    continue
                fn = getattr(cls, name)

                def wrap(fn=fn, name=name):
                    def w(self, *a, **k):
                        counter[name] += 1
                        return fn(self, *a, **k)
                    return w
                setattr(cls, name, wrap())
            return devc
        return f

    counter = collections.Counter()
    bt.cpu_numpy = counting(cpu_o1)
    dev = rapi.Device("cpu")
    np.random.seed(0)
    model = RefMLP()
    opt = roptim.SGD(model.parameters(), lr=0.05)
    crit = rnn.CrossEntropyLoss()
    xb = np.random.rand(128, 784).astype(np.float32); yb = np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 128)]
    for step in range(2):
        counter.clear()
        out = model(RT(xb, device=dev)); loss = crit(out, RT(yb, device=dev))
        opt.zero_grad(); loss.backward(); opt.step(); rt.Graph.free_graph()
    hist["mlp"] = dict(counter)
    import json
    json.dump(hist, open(os.path.join(OUT, "call_histograms.json"), "w"), indent=1, sort_keys=True)
    bt.cpu_numpy = cpu_o1
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, "total bytes", total)


if __name__ == "__main__":
    main()
