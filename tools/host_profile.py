"""Host-side profile of one training step (cProfile over K steps, GPU queue drained first): where
the Python / ctypes time of the hot path goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepflows_amd import backend_api, workloads  # noqa: E402
from deepflows_amd.tensor import Graph  # noqa: E402

dev = backend_api.Device("hip")
model, opt, crit, xt, yt = bench.build_job("resnet18", dev, 0, 256, register_blocks=True)


def step():
    loss, _ = workloads.train_step(model, crit, opt, xt, yt, sync=None)
    Graph.free_graph()
    return loss


for _ in range(5):
    step()
dev.mod.synchronize()
K = 20
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step()
pr.disable()
dev.mod.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"per-step totals over {K} steps (divide by {K})")
st.print_stats(28)
