"""Round 5 probe: what the ResNet-18 step takes on the GPU when the backward chain and the grouped weight gradients run on
CU-partitioned streams (DFHIP_OVERLAP_CUS=n + DFHIP_WGRAD_SIDE=k), issued eagerly behind the profiling gate (every launch
of the step is queued before the first one runs, so the GPU sees them back to back as in a replayed graph).
Prints one line: ms per step between two events on the library stream (median / best of the steps)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepflows_amd import backend_api, workloads  # noqa: E402
from deepflows_amd.tensor import Graph  # noqa: E402


def main():
    dev = backend_api.Device("hip")
    mod = dev.mod
    model, opt, crit, xt, yt = bench.build_job("resnet18", dev, 0, 256, register_blocks=True, model_kind="plain")
    for _ in range(5):
        loss, _ = workloads.train_step(model, crit, opt, xt, yt)
        Graph.free_graph()
    mod.synchronize()
    times = []
    for _ in range(int(os.environ.get("STEPS", "30"))):
        e0, e1 = mod.event(), mod.event()
        mod.prof_gate_close(150)
        mod.event_record(e0)
        loss, _ = workloads.train_step(model, crit, opt, xt, yt)
        Graph.free_graph()
        mod.event_record(e1)
        mod.prof_gate_open()
        mod.synchronize()
        times.append(mod.event_elapsed_ms(e0, e1))
    times = np.array(times[5:])
    print(json.dumps({"overlap_cus": os.environ.get("DFHIP_OVERLAP_CUS"), "wgrad_side": os.environ.get("DFHIP_WGRAD_SIDE"),
                      "cu_mask": os.environ.get("DFHIP_CU_MASK"),
                      "median_ms": float(np.median(times)), "best_ms": float(times.min()), "loss": float(loss.numpy()[0])}))


if __name__ == "__main__":
    main()
