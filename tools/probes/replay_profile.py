#!/usr/bin/env python3
"""cProfile of CapturedTrainStep.replay() for a workload (host side only): tools/probes/replay_profile.py cnn"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from deepflows_amd import backend_api
from deepflows_amd.graph_step import CapturedTrainStep
from deepflows_amd import workloads
from deepflows_amd.tensor import Graph
wl = sys.argv[1] if len(sys.argv) > 1 else "cnn"
dev = backend_api.Device("hip")
model, opt, crit, xt, yt = bench.build_job(wl, dev, 0, bench.WORKLOADS[wl]["batch"])
for _ in range(3):
    workloads.train_step(model, crit, opt, xt, yt); Graph.free_graph()
cap = CapturedTrainStep(model, crit, opt, xt, yt)
for _ in range(3): cap.replay()
dev.mod.synchronize()
K = 50
pr = cProfile.Profile(); pr.enable()
for _ in range(K): cap.replay()
pr.disable(); dev.mod.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime")
print(f"totals over {K} replays"); st.print_stats(14)
