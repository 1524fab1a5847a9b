#!/usr/bin/env python3
"""bench.py — training-step images/sec of the reference's ResNet-18 CIFAR-10 workload
(test/ResNet_CIFAR10_cuda.py: bs 256, fp32, Adam lr 1e-3 wd 5e-4, label-smoothed one-hot
targets) on the `hip` device, 1..8 MI355X (one process per GPU, weak scaling, gradients
all-reduced over RCCL).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = forward + loss + zero_grad + backward + (bucket all-reduce) + Adam step on one
batch of 256 synthetic images per GPU that is already resident in HBM.  Rank 0 prints ONE
JSON line (contract in the task statement) including
  roofline      live HIP-event timing of the dominant kernel family (events on the library
                stream around every launch of each family, in a second pass over the same K
                steps so the event records do not perturb the headline timing);
  cpu_baseline  the NumPy oracle (reference algorithm, primitive path) timed on this host on a
                bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 256
PEAK_MFMA_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak


def make_batch(rank, batch=BATCH):
    """SURVEY §8d synthetic inputs: default_rng(1234+rank); images N(0,1) clipped to [-1,1];
    labels uniform -> one-hot with eps = 0.05 smoothing."""
    rng = np.random.default_rng(1234 + rank)
    x = np.clip(rng.standard_normal((batch, 3, 32, 32)).astype(np.float32), -1.0, 1.0)
    labels = rng.integers(0, 10, batch)
    y = np.eye(10, dtype=np.float32)[labels] * 0.95 + 0.05 / 10
    return x, y.astype(np.float32)


def build_job(dev, rank, batch=BATCH, register_blocks=True):
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Tensor
    np.random.seed(0)                      # identical replicas on every rank
    model = workloads.ResNet18(10, device=dev, register_blocks=register_blocks)
    model.train()
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
    crit = nn.CrossEntropyLoss()
    x, y = make_batch(rank, batch)
    return model, opt, crit, Tensor(x, device=dev), Tensor(y, device=dev)


def cpu_baseline(seconds_budget=25.0):
    """The oracle on this host: NumPy device + the front-end's primitive path = the
    reference's algorithm (im2col by k*k strided copies + np.matmul, unfused BN / CE / Adam).
    Bounded sample: ResNet-18 CIFAR step at batch 32 (1/8 of the GPU batch)."""
    from deepflows_amd import backend_api, workloads
    from deepflows_amd.tensor import Graph
    from oracle.numpy_device import NumpyDevice
    dev = backend_api.register_device("cpu", NumpyDevice())
    try:
        sample_batch = 32
        model, opt, crit, xt, yt = build_job(dev, 0, sample_batch)
        t_first = time.perf_counter()
        workloads.train_step(model, crit, opt, xt, yt)
        Graph.free_graph()
        first = time.perf_counter() - t_first
        n = int(max(1, min(12, (seconds_budget - first) // max(first, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n):
            workloads.train_step(model, crit, opt, xt, yt)
            Graph.free_graph()
        dt = (time.perf_counter() - t0) / n
        Graph.free_graph_all()
    finally:
        backend_api.unregister_device("cpu")
    threads = int(os.environ.get("OPENBLAS_NUM_THREADS") or os.environ.get("OMP_NUM_THREADS") or
                  (os.cpu_count() or 1))
    return {"value": sample_batch / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"ResNet-18 CIFAR-10 train step at batch {sample_batch} (1/8 of the GPU batch), "
                      f"{n} timed step(s) after 1 warm-up, {dt:.2f} s/step, NumPy {np.__version__}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dgrad", choices=["overwrite", "accumulate"], default="overwrite",
                    help="conv input-gradient mode: reference-exact (default) or mathematically correct")
    ap.add_argument("--registration", choices=["all", "f6"], default="all",
                    help="all = every block registered (62 tensors); f6 = the reference script's 5 tensors")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the step from a hipGraph")
    ap.add_argument("--overlap", action="store_true",
                    help="data parallel: two graphs, the all-reduce of layer4 + fc beside the rest of the backward "
                         "pass (costs ~0.1 ms of split / event overhead per step: only pays for slow links)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")

    import deepflows_amd as df
    from deepflows_amd import backend_api, parallel, workloads
    from deepflows_amd.tensor import Graph

    dev = backend_api.Device("hip")            # raises when the extension / GPU is missing
    mod = dev.mod
    mod.conv_dgrad_mode = args.dgrad
    comm = parallel.init_communicator(dev)
    model, opt, crit, xt, yt = build_job(dev, rank, BATCH, register_blocks=(args.registration == "all"))
    bucket = parallel.GradientBucket(opt.params, comm)
    opt.grad_scale = 1.0 / comm.world
    # DF_FORCE_SYNC=1 (with DF_FORCE_COMM=1): exercise the exchange path of a multi-GPU run on one GPU
    sync = bucket.sync if (comm.world > 1 or os.environ.get("DF_FORCE_SYNC") == "1") else None

    def step():
        loss, _ = workloads.train_step(model, crit, opt, xt, yt, sync=sync)
        Graph.free_graph()
        return loss

    for _ in range(args.warmup):
        loss = step()
    mod.synchronize()

    # the timed steps replay one captured step (hipGraph) unless --no-graph; same kernels, same
    # order, one host call per step instead of ~170
    captured = None
    overlapped = False
    if not args.no_graph and sync is not None and args.overlap:
        # data parallel: two graphs split where layer4's gradients are complete, the all-reduce of
        # that bucket runs on the communication stream beside the rest of the backward pass
        try:
            from deepflows_amd.graph_step import OverlappedCapturedTrainStep
            late, early = workloads.late_and_early_parameters(model, opt.params)
            captured = OverlappedCapturedTrainStep(model, crit, opt, xt, yt,
                                                   parallel.GradientBucket(late, comm),
                                                   parallel.GradientBucket(early, comm))
            for _ in range(2):
                loss = captured.replay()
            mod.synchronize()
            overlapped = True
        except Exception as e:
            print(f"[bench] overlapped capture failed ({type(e).__name__}: {e}); single-bucket path", file=sys.stderr)
            captured = None
            opt.device_hyper = False
    if captured is None and not args.no_graph:
        try:
            from deepflows_amd.graph_step import CapturedTrainStep
            captured = CapturedTrainStep(model, crit, opt, xt, yt, sync=sync)
            for _ in range(2):
                loss = captured.replay()
            mod.synchronize()
        except Exception as e:                      # stay on the eager path, say so
            print(f"[bench] step capture failed ({type(e).__name__}: {e}); timing the eager path", file=sys.stderr)
            captured = None
    eager_step = step
    if captured is not None:
        step = captured.replay

    comm.barrier()
    mod.synchronize()
    e0, e1 = mod.event(), mod.event()
    t0 = time.perf_counter()
    mod.event_record(e0)
    for _ in range(args.steps):
        loss = step()
    mod.event_record(e1)
    host_issue = time.perf_counter() - t0        # the host has queued everything; the GPU may still run
    mod.synchronize()
    comm.barrier()
    wall = time.perf_counter() - t0
    wall = comm.max_float(wall)
    gpu_ms = mod.event_elapsed_ms(e0, e1)
    final_loss = float(loss.numpy()[0])

    # second pass over the same K steps with per-family HIP events (roofline); events cannot be
    # recorded inside a graph replay, so this pass issues the same step eagerly
    if captured is not None:
        opt.device_hyper = False
    mod.prof_reset()
    mod.prof_enable(True)
    for _ in range(args.steps):
        eager_step()
    mod.synchronize()
    mod.prof_enable(False)
    prof = mod.prof_read()
    comm.barrier()

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = BATCH * comm.world * args.steps / wall
        fams = {}
        for name, r in prof.items():
            if r["calls"] == 0 or r["ms"] <= 0:
                continue
            avg_ms = r["ms"] / r["calls"]
            ent = {"calls_per_step": r["calls"] / args.steps, "avg_us": avg_ms * 1e3,
                   "ms_per_step": r["ms"] / args.steps}
            if r["flops"] > 0:
                ent["tflops"] = r["flops"] / r["calls"] / (avg_ms * 1e-3) / 1e12
            ent["gbs"] = r["bytes"] / r["calls"] / (avg_ms * 1e-3) / 1e9
            fams[name] = ent
        dom = max(fams, key=lambda k: fams[k]["ms_per_step"]) if fams else None
        roofline = None
        if dom:
            d = fams[dom]
            if "tflops" in d:
                roofline = {"kernel": dom, "bound": "mfma", "achieved": d["tflops"], "peak": PEAK_MFMA_F32_TFLOPS,
                            "unit": "TFLOP/s", "frac": d["tflops"] / PEAK_MFMA_F32_TFLOPS, "traffic": None}
            else:
                roofline = {"kernel": dom, "bound": "hbm", "achieved": d["gbs"], "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": d["gbs"] / PEAK_HBM_GBS, "traffic": None}
            roofline["avg_launch_us"] = d["avg_us"]
            roofline["launches_per_step"] = d["calls_per_step"]
            # HBM bytes per launch of that family from the committed PMC passes (rocprofv3 --pmc
            # FETCH_SIZE / WRITE_SIZE on this same command, FETCH_SIZE x2 per the gfx950 note);
            # counters cannot be read from inside this process, so the number is the one on file
            try:
                import json as _json
                pmc = _json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                                   "r01p_pmc_traffic.json")))
                fam = pmc["families"].get(dom)
                if fam:
                    roofline["traffic"] = fam["hbm_bytes_per_step"] / d["calls_per_step"]
                    roofline["traffic_source"] = "profiles/r01p_pmc_traffic.json"
            except Exception:
                pass
            roofline["algorithmic_bytes_per_launch"] = prof[dom]["bytes"] / prof[dom]["calls"]
            roofline["algorithmic_flops_per_launch"] = prof[dom]["flops"] / prof[dom]["calls"]
        out = {
            "metric": "training-step images/sec, ResNet-18 CIFAR-10 bs=256",
            "value": value, "unit": "images/s", "n_gpus": comm.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ResNet-18 (CIFAR variant 32/64/128/256, test/ResNet_CIFAR10_cuda.py) "
                                   "train step: fwd + CE + bwd + Adam, bs 256 per GPU, fp32",
                       "global_batch": BATCH * comm.world, "parallelism": f"dp{comm.world}",
                       "conv_dgrad": args.dgrad, "dispatch": ("hipgraph+overlapped-allreduce" if overlapped else "hipgraph") if captured is not None else "eager", "registered_params": args.registration,
                       "allreduce_elems": bucket.total if comm.world > 1 else 0},
            "gpu_ms_per_step": gpu_ms / args.steps, "host_issue_ms_per_step": 1e3 * host_issue / args.steps,
            "final_loss": final_loss,
            "roofline": roofline,
            "kernels": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in
                        sorted(fams.items(), key=lambda kv: -kv[1]["ms_per_step"])},
        }
        if comm.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        line = json.dumps(out)
    else:
        line = None
    comm.barrier()
    comm.close()
    # RCCL writes a version banner to the C stdout buffer, which is flushed at process exit — after
    # anything Python printed.  Drain it first so that the JSON line is the last line of stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
