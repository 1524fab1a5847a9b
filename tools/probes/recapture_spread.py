"""Is the process-to-process spread of the step time a property of the process or of the buffers a capture got?  One process:
build the job, capture, time 3000 replays, destroy -- four times with the pool's blocks reused, four times with the pool trimmed
in between (fresh hipMalloc blocks)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from deepflows_amd import backend_api, workloads
from deepflows_amd.graph_step import CapturedTrainStep
from deepflows_amd.tensor import Graph

dev = backend_api.Device("hip")
mod = dev.mod
for trim in (False, True):
    out = []
    for it in range(4):
        np.random.seed(0)
        model, opt, crit, xt, yt = bench.build_job("resnet18", dev, 0, 256)
        for _ in range(2):
            workloads.train_step(model, crit, opt, xt, yt); Graph.free_graph()
        cap = CapturedTrainStep(model, crit, opt, xt, yt)
        for _ in range(300):
            cap.replay()
        mod.synchronize()
        t0 = time.perf_counter()
        for _ in range(3000):
            cap.replay()
        mod.synchronize()
        out.append((time.perf_counter() - t0) / 3000 * 1e3)
        cap.destroy(); Graph.free_graph_all()
        del model, opt, crit, xt, yt, cap
        if trim:
            import gc; gc.collect()
            mod.pool_trim() if hasattr(mod, "pool_trim") else None
    print("pool trimmed between captures" if trim else "pool blocks reused", " ".join(f"{v:.4f}" for v in out))
