#!/usr/bin/env python3
"""Step time of the other two reference workloads: MLP 784-256-10 bs 128 (SGD)
and the CIFAR-10 CNN bs 256 (Adam), eager, on the hip device."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepflows_amd import backend_api, nn, optim, workloads  # noqa: E402
from deepflows_amd.tensor import Graph, Tensor  # noqa: E402

dev = backend_api.Device("hip")
mod = dev.mod
rng = np.random.default_rng(0)


def run(name, model, opt, x, y, steps=30):
    crit = nn.CrossEntropyLoss()
    xt, yt = Tensor(x, device=dev), Tensor(y, device=dev)
    for _ in range(5):
        workloads.train_step(model, crit, opt, xt, yt)
        Graph.free_graph()
    mod.synchronize()
    mod.prof_reset(); mod.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        workloads.train_step(model, crit, opt, xt, yt)
        Graph.free_graph()
    mod.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mod.prof_enable(False)
    prof = mod.prof_read()
    top = sorted(((r["ms"] / steps, k) for k, r in prof.items() if r["calls"]), reverse=True)[:6]
    print(f"{name}: {dt * 1e3:.3f} ms/step, {x.shape[0] / dt:.0f} samples/s; top families (ms/step):",
          ", ".join(f"{k} {v:.3f}" for v, k in top))


np.random.seed(0)
mlp = workloads.MLP((784, 256, 10), device=dev)
if mlp is not None:
    run("MLP bs128", mlp, optim.SGD(mlp.parameters(), lr=0.05), rng.random((128, 784)).astype(np.float32),
        np.eye(10, dtype=np.float32)[rng.integers(0, 10, 128)])
np.random.seed(0)
cnn = workloads.CIFAR10_CNN(10, device=dev)
run("CNN bs256", cnn, optim.Adam(cnn.parameters(), lr=5e-3, weight_decay=5e-4),
    np.clip(rng.standard_normal((256, 3, 32, 32)), -1, 1).astype(np.float32),
    np.eye(10, dtype=np.float32)[rng.integers(0, 10, 256)])
