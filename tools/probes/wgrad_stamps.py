#!/usr/bin/env python3
"""Per-layer / per-workgroup cycle stamps of the grouped weight gradient inside one eager ResNet-18 step (diagnostic build:
ONLY=dfhip_wgrad tools/build_variant.sh wstamps -DDFHIP_STAMPS;
DFHIP_LIBRARY=tools/probes/build/wstamps/libdfhip.so DFHIP_CONV_STAMP=1 python3 tools/probes/wgrad_stamps.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from deepflows_amd import backend_api, workloads
from deepflows_amd.tensor import Graph
dev = backend_api.Device("hip")
dev.mod.conv_matmul_dtype = os.environ.get("MM", "f32")
model, opt, crit, xt, yt = bench.build_job(os.environ.get("WORKLOAD", "resnet18"), dev, 0, 256)
for i in range(int(os.environ.get("STEPS", "6"))):
    workloads.train_step(model, crit, opt, xt, yt)
    Graph.free_graph()
    dev.mod.synchronize()
    sys.stderr.write(f"---- step {i}\n")
