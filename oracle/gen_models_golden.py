#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only): golden vectors for the two other convolutional models
of the reference's example scripts -- SURVEY §8(f) row 3 -- produced by the reference ITSELF.

  test/VGG.py:7-152        VGG16Model  (13 conv+BN+ReLU, 5 max-pools, Linear 512-4096-4096-10, Dropout)
  test/MobileNet.py:18-108 ConvBlock / MobileNetV1 (plain 3x3 conv blocks, width_multiplier 0.25, SGD
                           lr .01 momentum .9 wd 5e-4 as in :158-160)

The classes are lifted from the scripts with `ast` (the scripts load datasets at import time) and
run on the reference's NumPy device with the scatter-correct setitem patch (O2, see gen_golden.py).
One training step at batch 2 each; inputs are regenerated from the recorded seed, the fixture holds
loss, logits, one weight gradient and per-parameter checksums -> tests/golden/g10_models.npz.
Usage (needs /root/reference):  python oracle/gen_models_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, import_reference, lift_classes  # noqa: E402


def main():
    bt, orig_cpu_numpy, cpu_numpy_o2 = import_reference()
    import DeepFlows
    from DeepFlows import nn as rnn, tensor as rt
    from DeepFlows.tensor import Tensor as RT
    from DeepFlows.optim.sgd import SGD
    from DeepFlows.optim.adam import Adam
    for mod in (bt, DeepFlows.backend, DeepFlows.backend_selection, DeepFlows.tensor):
        if hasattr(mod, "cpu_numpy"):
            mod.cpu_numpy = cpu_numpy_o2
    dev = DeepFlows.backend_api.Device("cpu")
    g = {}

    drng = np.random.default_rng(4321)
    xb = np.clip(drng.standard_normal((2, 3, 32, 32)).astype(np.float32), -1, 1)
    yb = np.eye(10, dtype=np.float32)[drng.integers(0, 10, 2)]
    g["x"], g["y"] = xb, yb

    def record(tag, model, opt, first_conv):
        params = list(model.parameters())
        g[f"{tag}_n_params"] = np.array([len(params), sum(int(np.prod(p.shape)) for p in params)])
        g[f"{tag}_init_abs_sum"] = np.array([float(np.abs(p.data.numpy()).sum()) for p in params])
        out = model(RT(xb, device=dev))
        loss = rnn.CrossEntropyLoss()(out, RT(yb, device=dev))
        opt.zero_grad()
        loss.backward()
        g[f"{tag}_loss"], g[f"{tag}_logits"] = loss.numpy(), out.numpy()
        g[f"{tag}_grad_abs_sum"] = np.array([float(np.abs(p.grad.numpy()).sum()) for p in params])
        g[f"{tag}_grad_sum"] = np.array([float(p.grad.numpy().sum()) for p in params])
        g[f"{tag}_conv1_wgrad"] = first_conv.weight.grad.numpy()
        opt.step()
        g[f"{tag}_param_abs_sum_after"] = np.array([float(np.abs(p.data.numpy()).sum()) for p in params])
        rt.Graph.free_graph_all()

    # VGG16 (test/VGG.py), 32x32 input, Adam as in the CNN / ResNet scripts
    ns = {"nn": rnn, "Tensor": RT, "__name__": "lifted"}
    exec(lift_classes("VGG.py", {"VGG16Model", "VGG16"}), ns)
    np.random.seed(0)
    model = ns["VGG16"](10, device="cpu", img_size=(32, 32))
    model.train()
    record("vgg", model, Adam(model.parameters(), lr=1e-3, weight_decay=5e-4), model.conv1_1)

    # MobileNetV1 (test/MobileNet.py), width 0.25, SGD momentum (test/MobileNet.py:158-160)
    ns = {"nn": rnn, "tensor": rt, "Tensor": RT, "Module": rnn.Module, "Conv2d": rnn.Conv2d,
          "BatchNorm2d": rnn.BatchNorm2d, "Linear": rnn.Linear, "__name__": "lifted"}
    code = lift_classes("MobileNet.py", {"ConvBlock", "make_divisible", "MobileNetV1"},
                        replace=[("from DeepFlows import maximum", "from DeepFlows.tensor import maximum")])
    exec(code, ns)
    np.random.seed(0)
    model = ns["MobileNetV1"](num_classes=10, width_multiplier=0.25, device="cpu", img_size=(32, 32))
    model.train()
    record("mbn", model, SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0005), model.conv1)

    # the repository's other ResNet-18 (test/ResNet.py:24-154: 64/128/256/512 channels, no stem pool, ReLU after the
    # residual add), SGD lr .01 momentum .9 wd 5e-4 as in :237.  Its blocks sit in Python lists, so the optimizer
    # sees 5 tensors; gradients are recorded for all 62.
    ns = {"nn": rnn, "tensor": rt, "Tensor": RT, "__name__": "lifted"}
    exec(lift_classes("ResNet.py", {"ResidualBlock", "ResNet", "ResNet18"}), ns)
    np.random.seed(0)
    model = ns["ResNet18"](10, img_size=(32, 32), device="cpu")
    model.train()
    opt = SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0005)

    def all_params(m):
        out = [m.conv1.weight, m.bn1.weight, m.bn1.bias]
        for stage in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in stage:
                if blk.downsample is not None:          # created before the block's own layers (:88-91)
                    out += [blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias]
                out += [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias]
        return out + [m.fc.weight, m.fc.bias]
    every = all_params(model)
    g["wide_n_params"] = np.array([len(every), sum(int(np.prod(p.shape)) for p in every), len(list(model.parameters()))])
    g["wide_init_abs_sum"] = np.array([float(np.abs(p.data.numpy()).sum()) for p in every])
    out = model(RT(xb, device=dev))
    loss = rnn.CrossEntropyLoss()(out, RT(yb, device=dev))
    opt.zero_grad()
    loss.backward()
    g["wide_loss"], g["wide_logits"] = loss.numpy(), out.numpy()
    g["wide_grad_abs_sum"] = np.array([float(np.abs(p.grad.numpy().astype(np.float64)).sum()) for p in every])
    g["wide_conv1_wgrad"] = model.conv1.weight.grad.numpy()
    g["wide_fc_wgrad"] = model.fc.weight.grad.numpy()
    opt.step()
    g["wide_param_abs_sum_after"] = np.array([float(np.abs(p.data.numpy().astype(np.float64)).sum()) for p in every])
    g["wide_running_var_bn1"] = model.bn1.running_var.numpy()
    rt.Graph.free_graph_all()

    np.savez_compressed(os.path.join(OUT, "g10_models.npz"), **g)
    print("wrote g10_models.npz:", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
