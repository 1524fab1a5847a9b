"""Host-side cost of one replayed CNN step, by part: is the replayed step bound by the host or by the device?
Prints the host time of N replays before the closing synchronize (the host running ahead of the device), the
whole time, and the host time of the parts of replay()."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from deepflows_amd import backend_api, workloads
from deepflows_amd.graph_step import CapturedTrainStep
from deepflows_amd.tensor import Graph

dev = backend_api.Device("hip")
mod = dev.mod
model, opt, crit, xt, yt = bench.build_job("cnn", dev, 0, 256)
for _ in range(3):
    workloads.train_step(model, crit, opt, xt, yt)
    Graph.free_graph()
cs = CapturedTrainStep(model, crit, opt, xt, yt)
for _ in range(20):
    cs.replay()
mod.synchronize()
N = 300
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N):
        cs.replay()
    t1 = time.perf_counter()
    mod.synchronize()
    t2 = time.perf_counter()
    print("replays: host %.1f us per step enqueued, %.1f us per step finished" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
# parts
def timeit(f, n=300):
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6
mod.synchronize()
print("push_hyper      %.1f us" % timeit(opt.push_hyper)); mod.synchronize()
print("masks (batch)   %.1f us" % timeit(lambda: mod.upload_bernoulli_masks(cs.feeds))); mod.synchronize()
print("get/set state   %.1f us" % timeit(lambda: np.random.set_state(np.random.get_state())))
print("graph.replay    %.1f us" % timeit(lambda: (cs.graph.replay(), opt.note_replayed_apply()))); mod.synchronize()
os.environ["DF_DEBUG_SKIP_FEEDS"] = "1"
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(N):
        cs.replay()
    t1 = time.perf_counter()
    mod.synchronize()
    t2 = time.perf_counter()
    print("no feeds: host %.1f us per step enqueued, %.1f us per step finished" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
