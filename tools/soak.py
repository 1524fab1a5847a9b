"""Soak run of the benchmarked step (run as `python3 tools/soak.py [--replays N]`; under rocprofv3 the same after `--`):

  * eager steps on a fixed batch: the loss falls, the pool stops growing after the first steps;
  * N (default 5 000) replays of the captured step of bench.py's headline job (ResNet-18, batch 256): the SUSTAINED time per
    step next to the 30-replay number bench.py times, with the clock the chip holds under a dense fp32-MFMA load before and
    after (dfhip_clock_probe) -- one JSON line, kept under profiles/ per round."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepflows_amd import backend_api, workloads  # noqa: E402
from deepflows_amd.graph_step import CapturedTrainStep  # noqa: E402
from deepflows_amd.tensor import Graph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replays", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", choices=["plain", "fused"], default="plain")
    args = ap.parse_args()
    dev = backend_api.Device("hip")
    mod = dev.mod
    model, opt, crit, xt, yt = bench.build_job("resnet18", dev, 0, args.batch, register_blocks=True, model_kind=args.model)
    losses = []
    for i in range(30):
        loss, _ = workloads.train_step(model, crit, opt, xt, yt)
        Graph.free_graph()
        losses.append(float(loss.numpy()[0]))
    s0 = mod.pool_stats()
    for i in range(70):
        loss, _ = workloads.train_step(model, crit, opt, xt, yt)
        Graph.free_graph()
    losses.append(float(loss.numpy()[0]))
    s1 = mod.pool_stats()
    print("eager: loss %.4f -> %.4f -> %.4f; hipMalloc calls %d -> %d, reserved %.1f MB" %
          (losses[0], losses[29], losses[-1], s0["hipmalloc_calls"], s1["hipmalloc_calls"], s1["reserved_bytes"] / 1e6),
          file=sys.stderr)
    assert losses[-1] < losses[0] and np.isfinite(losses[-1])
    assert s1["hipmalloc_calls"] == s0["hipmalloc_calls"], "the pool kept growing in the steady state"
    cs = CapturedTrainStep(model, crit, opt, xt, yt)
    for _ in range(2):
        cs.replay()
    mod.synchronize()
    clock0 = mod.clock_probe(200.0)
    s1b = mod.pool_stats()      # (the capture itself and the clock probe's buffer are behind us: from here on, replays only)
    # the 30-replay region bench.py times, then the long run in chunks (the per-chunk times show a drift, if any)
    t0 = time.perf_counter()
    for _ in range(30):
        loss = cs.replay()
    mod.synchronize()
    short_ms = (time.perf_counter() - t0) * 1e3 / 30
    chunks = []
    n_chunk = 10
    per = max(1, args.replays // n_chunk)
    t_all = time.perf_counter()
    for _ in range(n_chunk):
        t0 = time.perf_counter()
        for _ in range(per):
            loss = cs.replay()
        mod.synchronize()
        chunks.append((time.perf_counter() - t0) * 1e3 / per)
    total = time.perf_counter() - t_all
    s2 = mod.pool_stats()
    clock1 = mod.clock_probe(200.0)
    lr = float(loss.numpy()[0])
    assert np.isfinite(lr) and lr < losses[-1] + 1e-3
    print(json.dumps({"job": "ResNet-18 CIFAR-10 bs=%d, captured step, model=%s" % (args.batch, args.model),
                      "replays": per * n_chunk, "ms_per_step_sustained": total * 1e3 / (per * n_chunk),
                      "images_per_s_sustained": args.batch * per * n_chunk / total,
                      "ms_per_step_first_30": short_ms, "ms_per_step_by_tenth": [round(c, 4) for c in chunks],
                      "mfma_clock_mhz_before": clock0[0], "mfma_clock_mhz_after": clock1[0],
                      "mfma_probe_tflops_before": clock0[1], "mfma_probe_tflops_after": clock1[1],
                      "final_loss": lr, "reserved_mb": s2["reserved_bytes"] / 1e6,
                      "hipmalloc_calls_capture_and_probe": s1b["hipmalloc_calls"] - s1["hipmalloc_calls"],
                      "hipmalloc_calls_during_replays": s2["hipmalloc_calls"] - s1b["hipmalloc_calls"]}))
    assert s2["hipmalloc_calls"] == s1b["hipmalloc_calls"], "a replayed step reached hipMalloc"


if __name__ == "__main__":
    main()
