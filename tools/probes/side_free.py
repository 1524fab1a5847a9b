"""What a long matrix-bound kernel on a SECOND queue costs the captured step when nothing but the hardware orders the two
(no fork / join edges inside a graph): per step one k_clock_probe launch of `wgs` workgroups x `iters` x 4 MFMAs per wave on
the side stream (each workgroup holds 150 KB of LDS: a CU of its own), then the replay of the captured ResNet-18 step on the
library stream.  Run with DFHIP_SIDE_FREE=1:   DFHIP_SIDE_FREE=1 python3 tools/probes/side_free.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from deepflows_amd import backend_api, workloads  # noqa: E402
from deepflows_amd.graph_step import CapturedTrainStep  # noqa: E402
from deepflows_amd.tensor import Graph  # noqa: E402

dev = backend_api.Device("hip")
mod = dev.mod
model, opt, crit, xt, yt = bench.build_job("resnet18", dev, 0, 256)
for _ in range(3):
    workloads.train_step(model, crit, opt, xt, yt)
    Graph.free_graph()
cs = CapturedTrainStep(model, crit, opt, xt, yt)
for _ in range(3):
    cs.replay()
mod.synchronize()


def run(n, wgs, iters):
    mod.side_join()
    mod.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if wgs:
            mod.side_probe(wgs, iters, 150000)
        cs.replay()
    mod.side_join()
    mod.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


# the side kernel alone: its duration
def alone(wgs, iters, n=50):
    mod.side_join()
    mod.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        mod.side_probe(wgs, iters, 150000)
    mod.side_join()
    mod.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


print("step alone                         %.4f ms" % run(200, 0, 0))
for wgs, iters in ((8, 100), (64, 2500), (128, 2500), (128, 5000), (128, 10000), (192, 5000)):
    print("side load %3d WGs x %5d iters: alone %.4f ms, step beside it %.4f ms" % (wgs, iters, alone(wgs, iters), run(200, wgs, iters)))
