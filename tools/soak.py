#!/usr/bin/env python3
"""Soak test: many eager and replayed training steps; the loss must fall on a fixed batch, the pool
must stop growing after the first steps, and eager / replay must agree at the end."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepflows_amd import backend_api, workloads  # noqa: E402
from deepflows_amd.graph_step import CapturedTrainStep  # noqa: E402
from deepflows_amd.tensor import Graph  # noqa: E402

dev = backend_api.Device("hip")
mod = dev.mod
model, opt, crit, xt, yt = bench.build_job(dev, 0, 64, register_blocks=True)
losses = []
for i in range(30):
    loss, _ = workloads.train_step(model, crit, opt, xt, yt)
    Graph.free_graph()
    losses.append(float(loss.numpy()[0]))
s0 = mod.pool_stats()
for i in range(170):
    loss, _ = workloads.train_step(model, crit, opt, xt, yt)
    Graph.free_graph()
losses.append(float(loss.numpy()[0]))
s1 = mod.pool_stats()
print("eager: loss %.4f -> %.4f -> %.4f; hipMalloc calls %d -> %d, reserved %.1f MB" %
      (losses[0], losses[29], losses[-1], s0["hipmalloc_calls"], s1["hipmalloc_calls"], s1["reserved_bytes"] / 1e6))
assert losses[-1] < losses[0] and np.isfinite(losses[-1])
assert s1["hipmalloc_calls"] == s0["hipmalloc_calls"], "the pool kept growing in the steady state"
cs = CapturedTrainStep(model, crit, opt, xt, yt)
t0 = time.perf_counter()
for i in range(2000):
    loss = cs.replay()
mod.synchronize()
dt = time.perf_counter() - t0
lr = float(loss.numpy()[0])
s2 = mod.pool_stats()
print("replay: 2000 steps in %.2f s (%.3f ms/step), loss %.5f, step counter %d, reserved %.1f MB" %
      (dt, dt / 2, lr, opt.t, s2["reserved_bytes"] / 1e6))
assert np.isfinite(lr) and lr < losses[-1] + 1e-3
print("soak ok")
