"""Captured replays against eager steps over many steps (default 300) for the CNN and the perceptron: the deferred launches of
round 6 (masks from one call, Dropout + classifier + loss, riders of the slab sum, the optimizer's step inside the perceptron's
backward launch) must leave bit-identical parameters whichever way the step is issued.  Prints one line per workload."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from deepflows_amd import backend_api, workloads
from deepflows_amd.graph_step import CapturedTrainStep
from deepflows_amd.tensor import Graph

steps = int(os.environ.get("STEPS", "300"))
dev = backend_api.Device("hip")
for wl, batch in (("cnn", 256), ("mlp", 128)):
    outs = []
    for captured in (False, True):
        np.random.seed(0)
        model, opt, crit, xt, yt = bench.build_job(wl, dev, 0, batch)
        np.random.seed(123)                       # the masks' stream
        losses = []
        if captured:
            cap = CapturedTrainStep(model, crit, opt, xt, yt)
            for _ in range(steps):
                loss = cap.replay()
            losses.append(float(loss.numpy()[0]))
            cap.destroy()
        else:
            for _ in range(steps):
                loss, _ = workloads.train_step(model, crit, opt, xt, yt)
                Graph.free_graph()
            losses.append(float(loss.numpy()[0]))
        outs.append((losses, [p.data.numpy().copy() for p in model.parameters()], np.random.random_sample(2)))
        Graph.free_graph_all()
    same = all(np.array_equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    print(f"{wl}: {steps} steps, final loss eager {outs[0][0][-1]:.6f} captured {outs[1][0][-1]:.6f}, parameters bit-identical: {same}, "
          f"generator position identical: {np.array_equal(outs[0][2], outs[1][2])}, finite: {all(np.isfinite(a).all() for a in outs[1][1])}")
