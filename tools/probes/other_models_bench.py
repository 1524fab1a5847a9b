"""Step time of the other models the reference's scripts build (test/VGG.py, test/MobileNet.py) on the hip
device, eager and as a captured step, with the six launch families that take the most time -- a check that no layer of them
falls onto a slow generic path."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from deepflows_amd import backend_api, nn, optim, workloads
from deepflows_amd.graph_step import CapturedTrainStep
from deepflows_amd.tensor import Graph, Tensor

dev = backend_api.Device("hip")
mod = dev.mod
rng = np.random.default_rng(0)
B = int(os.environ.get("BATCH", "128"))
x = np.clip(rng.standard_normal((B, 3, 32, 32)), -1, 1).astype(np.float32)
y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, B)]
for name, make, mkopt in (
        ("VGG16 (test/VGG.py), Adam", lambda: workloads.VGG16(10, device=dev, img_size=(32, 32)), lambda p: optim.Adam(p, lr=1e-3)),
        ("MobileNetV1 x1.0 (test/MobileNet.py), SGD momentum", lambda: workloads.MobileNetV1(10, 1.0, device=dev, img_size=(32, 32)),
         lambda p: optim.SGD(p, lr=0.05, momentum=0.9, weight_decay=5e-4))):
    np.random.seed(0)
    model = make(); model.train()
    opt = mkopt(model.parameters())
    crit = nn.CrossEntropyLoss()
    xt, yt = Tensor(x, device=dev), Tensor(y, device=dev)
    for _ in range(3):
        workloads.train_step(model, crit, opt, xt, yt); Graph.free_graph()
    mod.synchronize()
    mod.prof_reset(); mod.prof_enable(True)
    steps = 10
    for _ in range(steps):
        workloads.train_step(model, crit, opt, xt, yt); Graph.free_graph()
    mod.synchronize(); mod.prof_enable(False)
    prof = mod.prof_read()
    launches = sum(r["calls"] for r in prof.values()) / steps
    kern = sum(r["ms"] for r in prof.values()) / steps
    top = sorted(((r["ms"] / steps, r["calls"] / steps, k) for k, r in prof.items() if r["calls"]), reverse=True)[:7]
    cap = CapturedTrainStep(model, crit, opt, xt, yt)
    for _ in range(5):
        cap.replay()
    mod.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        cap.replay()
    mod.synchronize()
    dt = (time.perf_counter() - t0) / 50
    cap.destroy(); Graph.free_graph_all()
    print(f"{name}, batch {B}: captured step {dt * 1e3:.3f} ms ({B / dt:.0f} images/s); {launches:.0f} launches, kernels {kern:.3f} ms; top: "
          + ", ".join(f"{k} {c:.0f} x {v / max(c, 1) * 1e3:.1f} us" for v, c, k in top))
