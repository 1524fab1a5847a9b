import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from deepflows_amd import backend_api, workloads
from deepflows_amd.graph_step import CapturedTrainStep
from deepflows_amd.tensor import Graph
dev = backend_api.Device("hip"); mod = dev.mod
model, opt, crit, xt, yt = bench.build_job("resnet18", dev, 0, 256)
for _ in range(3):
    workloads.train_step(model, crit, opt, xt, yt); Graph.free_graph()
cs = CapturedTrainStep(model, crit, opt, xt, yt)
for _ in range(3): cs.replay()
small = mod.Array(1024)
def run(n, between):
    mod.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        between(); cs.replay()
    mod.synchronize(); return (time.perf_counter() - t0) * 1e3 / n
print("replay only            %.4f ms" % run(200, lambda: None))
print("1 small kernel between %.4f ms" % run(200, lambda: mod.fill(small, 1.0)))
print("3 small kernels between %.4f ms" % run(200, lambda: (mod.fill(small, 1.0), mod.fill(small, 2.0), mod.fill(small, 3.0))))
import time as _t
def host_busy():
    t=_t.perf_counter()
    while _t.perf_counter()-t < 0.0004: pass
print("0.4 ms of host work between %.4f ms" % run(200, host_busy))
print("0.4 ms host + 3 kernels %.4f ms" % run(200, lambda: (host_busy(), mod.fill(small, 1.0), mod.fill(small, 2.0), mod.fill(small, 3.0))))
