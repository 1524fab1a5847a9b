"""TEST INFRASTRUCTURE — golden vectors for SURVEY §8(f) row 3 (the remaining nn surface).  Runs ONLY in the
build container (reference at /root/reference): imports the reference's own Python and records inputs +
outputs of

  * avg_pool2d forward / backward (nn/functional.py:377-404), including the zero-padding case (padded zeros are
    averaged with the full k*k divisor) and overlapping windows (the backward goes through __im2col2d.grad_fn:
    last writer wins, functional.py:285-294) — tier O2 (scatter setitem) as for Conv2d.  The reference's
    function itself cannot run: line 402 calls `col.mean(1)`, a method its Tensor class does not have
    (AttributeError, recorded in the fixture as ap_reference_error).  The expected values are therefore the
    function's OWN steps executed with the reference's primitives — its __pad2d, its __im2col2d, its
    transpose / reshape — and the window mean it documents ("二维平均池化") written as sum(col, 1) / k^2
    (the reference's tensor.mean(axis) would divide by the TOTAL element count, backend_tensor.py:659-662);
  * the LeakyReLU / Tanh / Softmax / LogSoftmax modules (nn/modules/activation.py:63-179) forward / backward;
  * nll_loss (functional.py:76-88), both reductions, forward / backward;
  * one Adam step of the reference's ResNet-18 (test/ResNet_CIFAR10_cuda.py:20-100) with the 'partial' freezing
    strategy of test/ResNet18_parameter_freezing_test.py:100-216 (conv1, bn1, layer1, layer2 frozen; the
    optimizer over filter(requires_grad));
  * the learning-rate sequences of optim/scheduler.py:13-61 (StepLR, CosineAnnealingLR, WarmupCosineLR).

    python oracle/gen_f3_golden.py        -> tests/golden/g11_nn_surface.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402

OUT = gg.OUT


def main():
    bt, cpu_o1, cpu_o2 = gg.import_reference()
    from DeepFlows import tensor as rt
    from DeepFlows import nn as rnn
    from DeepFlows.nn import functional as RF
    from DeepFlows import optim as roptim
    from DeepFlows.optim import scheduler as rsched
    from DeepFlows.tensor import Tensor as RT
    from DeepFlows import backend_api as rapi
    from DeepFlows.nn.modules import activation as ract     # LogSoftmax is defined there but not exported by nn

    bt.cpu_numpy = cpu_o2
    dev = rapi.Device("cpu")

    def T(a, rg=False):
        return RT(np.asarray(a, dtype=np.float32), device=dev, requires_grad=rg)

    def backward_with(y, g):
        rt.sum(y * T(g)).backward()

    g = {}
    rng = np.random.default_rng(311)

    # ---- avg_pool2d
    for name, (shape, k, s, p) in {"ap_k2s2": ((2, 3, 6, 6), 2, 2, 0), "ap_k3s1p1": ((2, 2, 5, 5), 3, 1, 1),
                                   "ap_k3s2p1": ((1, 4, 8, 8), 3, 2, 1)}.items():
        x = rng.standard_normal(shape).astype(np.float32)
        xt = T(x, True)
        try:
            RF.avg_pool2d(T(x), k, s, p)
            g["ap_reference_error"] = np.array("")
        except Exception as e:                      # AttributeError: 'Reshape' object has no attribute 'mean'
            g["ap_reference_error"] = np.array(f"{type(e).__name__}: {e}")
        rt.Graph.free_graph_all()
        col = getattr(RF, "__im2col2d")(getattr(RF, "__pad2d")(xt, p), k, s)
        oh, ow = col.shape[-2:]
        col = col.transpose(0, 4, 5, 1, 2, 3).reshape(-1, k * k)
        y = (rt.sum(col, 1, True) * (1.0 / (k * k))).reshape(shape[0], oh, ow, shape[1]).transpose(0, 3, 1, 2)
        gy = rng.standard_normal(y.shape).astype(np.float32)
        backward_with(y, gy)
        g[name + "_x"], g[name + "_y"], g[name + "_gy"], g[name + "_dx"] = x, y.numpy(), gy, xt.grad.numpy()
        g[name + "_geo"] = np.array([k, s, p])
        rt.Graph.free_graph_all()

    # ---- activation modules
    x = rng.standard_normal((5, 7)).astype(np.float32)
    x[0, 0] = 0.0                                    # LeakyReLU at exactly 0
    gy = rng.standard_normal((5, 7)).astype(np.float32)
    g["act_x"], g["act_gy"] = x, gy
    for name, m in (("leaky", rnn.LeakyReLU(0.1)), ("tanh", rnn.Tanh()), ("softmax", rnn.Softmax(dim=1)),
                    ("logsoftmax", ract.LogSoftmax(dim=1))):
        xt = T(x, True)
        y = m(xt)
        backward_with(y, gy)
        g[name + "_y"], g[name + "_dx"] = y.numpy(), xt.grad.numpy()
        rt.Graph.free_graph_all()

    # ---- nll_loss on log-probabilities
    logp = np.log(np.exp(x) / np.exp(x).sum(1, keepdims=True)).astype(np.float32)
    tgt = np.eye(7, dtype=np.float32)[rng.integers(0, 7, 5)]
    g["nll_logp"], g["nll_target"] = logp, tgt
    for red in ("mean", "sum"):
        lt = T(logp, True)
        loss = RF.nll_loss(lt, T(tgt), red)
        loss.backward()
        g[f"nll_{red}"], g[f"nll_{red}_dlogp"] = loss.numpy(), lt.grad.numpy()
        rt.Graph.free_graph_all()

    # ---- ResNet-18 (CIFAR variant), 'partial' freezing, one Adam step at batch 2
    code = gg.lift_classes("ResNet_CIFAR10_cuda.py", {"ResidualBlock", "ResNet", "ResNet18"})
    glb = {"nn": rnn, "tensor": rt, "Tensor": RT, "__name__": "lifted"}
    exec(code, glb)                                  # class definitions only (lifted with ast)
    np.random.seed(0)
    model = glb["ResNet18"](10, device="cpu")
    model.train()

    def block_params(layer):
        out = []
        for blk in layer:
            out += list(blk.parameters())
        return out

    registered = list(model.parameters())
    all_params = list(model.conv1.parameters()) + list(model.bn1.parameters())
    for layer in (model.layer1, model.layer2, model.layer3, model.layer4):
        all_params += block_params(layer)
    all_params += list(model.fc.parameters())
    frozen = list(model.conv1.parameters()) + list(model.bn1.parameters()) + block_params(model.layer1) + \
        block_params(model.layer2)
    for p in all_params:
        p.requires_grad = True
    for p in frozen:
        p.requires_grad = False
    opt = roptim.Adam(list(filter(lambda p: p.requires_grad, all_params)), lr=1e-3, weight_decay=5e-4)
    drng = np.random.default_rng(77)
    xb = np.clip(drng.standard_normal((2, 3, 32, 32)).astype(np.float32), -1, 1)
    yb = (np.eye(10, dtype=np.float32)[drng.integers(0, 10, 2)] * 0.95 + 0.005).astype(np.float32)
    before = [float(np.abs(p.data.numpy()).sum()) for p in all_params]
    out = model(RT(xb, device=dev))
    loss = rnn.CrossEntropyLoss()(out, RT(yb, device=dev))
    opt.zero_grad()
    loss.backward()
    g["frz_x"], g["frz_y"] = xb, yb
    g["frz_loss"], g["frz_logits"] = loss.numpy(), out.numpy()
    g["frz_n_params"] = np.array([len(all_params), len(frozen), len(registered)])
    g["frz_has_grad"] = np.array([p.grad is not None for p in all_params])
    g["frz_grad_abs_sum"] = np.array([float(np.abs(p.grad.numpy()).sum()) if p.grad is not None else 0.0
                                      for p in all_params])
    opt.step()
    after = [float(np.abs(p.data.numpy()).sum()) for p in all_params]
    g["frz_abs_sum_before"], g["frz_abs_sum_after"] = np.array(before), np.array(after)
    g["frz_fc_w_after"] = model.fc.weight.data.numpy()
    rt.Graph.free_graph_all()

    # ---- learning-rate schedules (host scalars, float64)
    class Opt:
        def __init__(self, lr):
            self.lr = lr

    def run(sched, o, n=25):
        seq = []
        for _ in range(n):
            sched.step()
            seq.append(o.lr)
        return np.array(seq, dtype=np.float64)

    o = Opt(0.1); g["lr_step"] = run(rsched.StepLR(o, step_size=3, gamma=0.5), o)
    o = Opt(0.05); g["lr_cosine"] = run(rsched.CosineAnnealingLR(o, T_max=7, eta_min=1e-5), o)
    o = Opt(1e-3); g["lr_warmcos"] = run(rsched.WarmupCosineLR(o, warmup_epochs=2, T_max=10, eta_min=1e-5), o)
    o = Opt(7.0); g["lr_warmcos_explicit"] = run(rsched.WarmupCosineLR(o, warmup_epochs=4, T_max=6, base_lr=0.02,
                                                                       warmup_start_lr=1e-4, eta_min=0.0), o)
    o = Opt(0.3); g["lr_warmcos_nowarm"] = run(rsched.WarmupCosineLR(o, warmup_epochs=0, T_max=5), o)

    np.savez_compressed(os.path.join(OUT, "g11_nn_surface.npz"), **g)
    print("written", os.path.join(OUT, "g11_nn_surface.npz"), os.path.getsize(os.path.join(OUT, "g11_nn_surface.npz")))


if __name__ == "__main__":
    main()
