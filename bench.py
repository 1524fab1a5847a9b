#!/usr/bin/env python3
"""bench.py — training-step images/sec of the reference's workloads on the `hip` device, 1..8 MI355X (one
process per GPU, weak scaling, gradients all-reduced over RCCL).  Default = the BASELINE headline: ResNet-18
CIFAR-10 (test/ResNet_CIFAR10_cuda.py: bs 256, fp32, Adam lr 1e-3 wd 5e-4, label-smoothed one-hot targets).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload resnet18|cnn|mlp] [--dgrad overwrite|accumulate]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        (the launcher only spawns the ranks and sets RANK / WORLD_SIZE / MASTER_*: nothing here imports torch)

A "step" = forward + loss + zero_grad + backward + (bucket all-reduce) + optimizer step on one batch of synthetic
inputs per GPU that is already resident in HBM.  Rank 0 prints ONE JSON line (contract in the task statement) with
  roofline      live HIP-event timing of the dominant kernel family (events on the library stream around every
                launch of each family, in a second pass over the same K steps so the event records do not perturb
                the headline timing; each eagerly issued step is queued behind a closed gate and released at once,
                so the launches run back to back as in the replay); `kernels` carries the same per family, each
                against its own roofline;
  cpu_baseline  the NumPy oracle (reference algorithm, primitive path) timed on this host on a bounded sample of
                the same workload (N=1, rank 0 only).
"""
import argparse
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 (matrix)
PEAK_MFMA_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 (families named *_bf16: the opt-in weight-gradient mode)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak

# families whose roofline is HBM bandwidth although they run MFMAs: the reference-exact ("overwrite") input
# gradient keeps one tap per input pixel -- 1/k^2 of the GEMM -- and is bound by reading dY and writing dX
HBM_BOUND_FAMILIES = {"conv2d_dgrad_overwrite_igemm"}

WORKLOADS = {
    "resnet18": {"batch": 256, "metric": "training-step images/sec, ResNet-18 CIFAR-10 bs=256",
                 "text": "ResNet-18 (CIFAR variant 32/64/128/256, test/ResNet_CIFAR10_cuda.py) train step: "
                         "fwd + CE + bwd + Adam, bs 256 per GPU, fp32"},
    "cnn": {"batch": 256, "metric": "training-step images/sec, CNN-CIFAR10 bs=256",
            "text": "CNN-CIFAR10 (3 x Conv2d k5 + BN + pool, Dropout, test/CNN_CIFAR10_cuda.py:61-113) train step: "
                    "fwd + CE + bwd + Adam lr 5e-3 wd 5e-4, bs 256 per GPU, fp32"},
    # the repository's other ResNet-18 (SURVEY §8d "a useful stress shape"): never the headline
    "resnet18_wide": {"batch": 256, "metric": "training-step images/sec, ResNet-18 (test/ResNet.py, 64-512 channels) bs=256",
                      "cpu_batch": 16,
                      "text": "ResNet-18 of test/ResNet.py:24-154 (64/128/256/512 channels, 32x32 inputs, no stem pool, ReLU "
                              "after the residual add; 852 GFLOP per step) train step: fwd + CE + bwd + SGD lr .01 momentum "
                              ".9 wd 5e-4 (:237), bs 256 per GPU, fp32"},
    "mlp": {"batch": 128, "metric": "training-step samples/sec, MLP-MNIST 784-256-10 bs=128",
            "text": "MLP 784->256->10 (test/MLP_MNIST_cuda.py:37-59) train step: fwd + CE + bwd + SGD lr .05, "
                    "bs 128 per GPU, fp32"},
}


def make_batch(workload, rank, batch):
    """SURVEY §8d synthetic inputs: default_rng(1234+rank); images N(0,1) clipped to [-1,1] (MLP: uniform [0,1));
    labels uniform -> one-hot (ResNet-18: eps = 0.05 smoothing)."""
    rng = np.random.default_rng(1234 + rank)
    if workload == "mlp":
        x = rng.random((batch, 784), dtype=np.float32)
    else:
        x = np.clip(rng.standard_normal((batch, 3, 32, 32)).astype(np.float32), -1.0, 1.0)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, batch)]
    if workload == "resnet18":
        y = y * 0.95 + 0.05 / 10
    return x, y.astype(np.float32)


def build_job(workload, dev, rank, batch, register_blocks=True, model_kind="plain"):
    """model_kind (resnet18 only): "plain" = workloads.ResNet18Plain, whose forward() is the text of the reference's script
    (test/ResNet_CIFAR10_cuda.py:31-47, 84-99: module calls, `out + identity`, two tensor.mean) -- the fusions happen
    inside the modules (nn/lazy.py); "fused" = workloads.ResNet18, the forward() rewritten around the fused entry points."""
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Tensor
    np.random.seed(0)                      # identical replicas on every rank
    if workload == "resnet18":
        cls = workloads.ResNet18Plain if model_kind == "plain" else workloads.ResNet18
        model = cls(10, device=dev, register_blocks=register_blocks)
        opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
    elif workload == "resnet18_wide":
        model = workloads.ResNet18Wide(10, device=dev, register_blocks=register_blocks)
        opt = optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0005)
    elif workload == "cnn":
        model = workloads.CIFAR10_CNN(10, device=dev)
        opt = optim.Adam(model.parameters(), lr=5e-3, weight_decay=5e-4)
    else:
        model = workloads.MLP((784, 256, 10), device=dev)
        opt = optim.SGD(model.parameters(), lr=0.05)
    model.train()
    crit = nn.CrossEntropyLoss()
    # replicas are identical (seed 0 above); what a rank draws AFTER the init -- Dropout masks -- is rank-local (SURVEY §8e
    # "seed + rank"): rank 0 continues the stream (= the single-GPU run, bit for bit), rank r > 0 re-seeds with seed + r
    from deepflows_amd import parallel
    parallel.seed_rank_local(0, rank)
    x, y = make_batch(workload, rank, batch)
    return model, opt, crit, Tensor(x, device=dev), Tensor(y, device=dev)


OTHER_CONFIGURATIONS = (("cnn_cifar10_bs256", "cnn", "overwrite", "f32"), ("mlp_mnist_bs128", "mlp", "overwrite", "f32"),
                        # the perceptron with FOUR steps per captured graph (the same kernels; a graph launch is a fifth of its
                        # 19-us step -- graph_step.CapturedTrainStep(steps_per_graph=4))
                        ("mlp_mnist_bs128_4_steps_per_graph", "mlp", "overwrite", "f32", ("--steps-per-graph", "4")),
                        ("resnet18_bs256_accumulate_dgrad", "resnet18", "accumulate", "f32"),
                        # opt-in reduced precision: bf16 operands in the layer1-2 forward convolutions' and the weight
                        # gradients' matrix products
                        ("resnet18_bs256_bf16_contractions", "resnet18", "overwrite", "bf16"))


def other_configurations(steps, timeout_s=150.0):
    """The BASELINE's other single-GPU configurations inside the default run (so that the driver's own run times them too, not
    only the builder's profiles/): CNN-CIFAR10 bs 256 and MLP-MNIST bs 128 (BASELINE.json configs 2-3), and the headline model
    in the mathematically complete input-gradient mode.  Each one is this same script in a CHILD PROCESS (subprocess: fork +
    exec of a fresh interpreter, never an exec of this one) under a timeout, started after the headline has been measured: a
    fault, abort or hang in one of them costs that entry (reported as `error`) and nothing else.  The child runs
    `--steps` timed replays of its captured step and the per-family event pass; what comes back is its JSON line."""
    out = {}
    for name, wl, dgrad, mm, *extra in OTHER_CONFIGURATIONS:
        k = steps if wl != "mlp" else max(steps, 200)
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--dgrad", dgrad, "--matmul-dtype", mm,
               "--steps", str(k), "--warmup", "4",
               "--no-cpu-baseline", "--no-other-configs", "--no-other-model", "--sustain-ms", "0", "--settle-ms", "100"]
        cmd += list(extra[0]) if extra else []
        env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        t = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"child exited {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
                continue
            d = json.loads(lines[-1])
            rf = d.get("roofline") or {}
            out[name] = {"value": d["value"], "unit": "samples/s" if wl == "mlp" else "images/s",
                         "ms_per_step": d["ms_per_step"], "steps": d["steps"], "final_loss": d.get("final_loss"),
                         "conv_dgrad": dgrad, "conv_matmul_dtype": mm, "dispatch": d["config"].get("dispatch"),
                         "dominant_family": rf.get("kernel"), "roofline_bound": rf.get("bound"),
                         "roofline_frac": rf.get("frac"), "avg_launch_us": rf.get("avg_launch_us"),
                         "launches_per_step": d["config"].get("launches_per_step"),
                         "steps_per_graph": d["config"].get("steps_per_graph"),
                         "child_wall_s": round(time.perf_counter() - t, 1)}
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"child killed after {timeout_s:.0f} s"}
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def host_cores():
    """(physical cores of one socket, logical CPUs) from lscpu; falls back to os.cpu_count()."""
    logical = os.cpu_count() or 1
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        vals = {}
        for line in out.splitlines():
            k, _, v = line.partition(":")
            vals[k.strip()] = v.strip()
        per_socket = int(vals.get("Core(s) per socket", "0"))
        if per_socket > 0:
            return per_socket, logical
    except Exception:
        pass
    return logical, logical


def _oracle_steps(workload, batch, threads, max_steps, seconds_budget):
    """(seconds per step, timed steps) of the oracle on `threads` BLAS/OpenMP threads."""
    from threadpoolctl import threadpool_limits
    from deepflows_amd import backend_api, workloads
    from deepflows_amd.tensor import Graph
    from oracle.numpy_device import NumpyDevice
    dev = backend_api.register_device("cpu", NumpyDevice())
    try:
        with threadpool_limits(limits=threads):
            model, opt, crit, xt, yt = build_job(workload, dev, 0, batch)
            t_first = time.perf_counter()
            workloads.train_step(model, crit, opt, xt, yt)
            Graph.free_graph()
            first = time.perf_counter() - t_first
            n = int(max(1, min(max_steps, (seconds_budget - first) // max(first, 1e-3))))
            t0 = time.perf_counter()
            for _ in range(n):
                workloads.train_step(model, crit, opt, xt, yt)
                Graph.free_graph()
            dt = (time.perf_counter() - t0) / n
            Graph.free_graph_all()
    finally:
        backend_api.unregister_device("cpu")
    return dt, n


def cpu_baseline(workload, batch):
    """The oracle on this host: NumPy device + the front end's primitive path = the reference's algorithm (im2col by
    k*k strided copies + np.matmul, unfused BN / CE / optimizer).  Bounded sample of the SAME workload at the GPU's
    batch: 1 warm-up + 1-2 timed steps on the physical cores of one socket and again on 8 threads (the
    thread count of SURVEY §6's in-container figures; the oracle's strided copies do not scale with cores, so fewer
    threads can be faster).  `value` is the BETTER of the two configurations, `cores` the threads it used; both runs
    are listed under `runs`."""
    cores, logical = host_cores()
    mlp = workload == "mlp"
    gpu_batch = batch
    batch = WORKLOADS[workload].get("cpu_batch", batch)      # (the 852-GFLOP stress shape: a smaller batch keeps it bounded)
    runs = []
    for threads in sorted({cores, min(8, logical)}, reverse=True):
        dt, n = _oracle_steps(workload, batch, threads, 50 if mlp else 2, 3.0 if mlp else 14.0)
        runs.append({"threads": threads, "value": batch / dt, "s_per_step": dt, "timed_steps": n})
    best = max(runs, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "images/s", "cores": best["threads"], "kind": "port",
            "sample": f"{WORKLOADS[workload]['text'].split(' train step')[0]} train step at batch {batch} ("
                      f"{'the GPU batch' if batch == gpu_batch else f'the GPU runs batch {gpu_batch}'}), "
                      f"{best['timed_steps']} timed step(s) after 1 warm-up, {best['s_per_step']:.2f} s/step on "
                      f"{best['threads']} threads (best of {[r['threads'] for r in runs]} threads; host: {cores} "
                      f"physical cores per socket, {logical} logical CPUs), NumPy {np.__version__}",
            "runs": runs}


def csrc_fingerprint():
    """sha256 over the kernel sources: a PMC traffic file is only quoted for the build it was measured on."""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "deepflows_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "deepflows_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def pmc_file(workload, dgrad, explicit=None):
    """The PMC summary (tools/pmc_summary.py --json) to quote: `explicit` if given, otherwise the newest committed
    profiles/r*_pmc_traffic*.json that was taken on THIS build of the kernels, this workload and this mode.  Returns
    (dict or None, why / where)."""
    paths = [explicit] if explicit else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True)
    if not paths:
        return None, "no PMC pass on file"
    why = []
    for path in paths:
        rel = os.path.relpath(path, ROOT)
        try:
            pmc = json.load(open(path))
        except Exception as e:
            why.append(f"{rel} unreadable: {e}")
            continue
        if pmc.get("workload", "resnet18") != workload or pmc.get("dgrad", "overwrite") != dgrad:
            continue
        if pmc.get("csrc_sha16") != csrc_fingerprint():
            why.append(f"{rel} was measured on kernel sources {pmc.get('csrc_sha16')} (git {pmc.get('git', '?')}), this build "
                       f"is {csrc_fingerprint()}: stale, not quoted")
            continue
        return pmc, f"{rel} (git {pmc.get('git', '?')}, kernel sources {pmc['csrc_sha16']})"
    if len(why) > 2:                      # (the newest two say it all)
        why = why[:2] + [f"... and {len(why) - 2} older set(s)"]
    return None, "; ".join(why) if why else "no PMC pass of this workload / mode on file"


def family_rooflines(prof, steps):
    fams = {}
    for name, r in prof.items():
        if r["calls"] == 0 or r["ms"] <= 0:
            continue
        # kernel_ms: event pairs bound to the family's dispatches (the kernels' own start / end stamps, as rocprofv3's
        # kernel trace reports them); ms: stream events around the family call, i.e. the same plus the dispatch gap
        kernel_timed = r.get("kernel_launches", 0) > 0 and r.get("kernel_ms", 0.0) > 0
        tot_ms = r["kernel_ms"] if kernel_timed else r["ms"]
        avg_s = tot_ms / r["calls"] * 1e-3
        ent = {"calls_per_step": r["calls"] / steps, "avg_us": avg_s * 1e6, "ms_per_step": tot_ms / steps,
               "avg_us_with_dispatch": r["ms"] / r["calls"] * 1e3,
               "kernels_per_call": (r["kernel_launches"] / r["calls"]) if kernel_timed else None,
               "gbs": r["bytes"] / r["calls"] / avg_s / 1e9}
        if r["flops"] > 0:
            ent["tflops"] = r["flops"] / r["calls"] / avg_s / 1e12
        if r["flops"] > 0 and name not in HBM_BOUND_FAMILIES:
            peak = PEAK_MFMA_BF16_TFLOPS if name.endswith("_bf16") else PEAK_MFMA_F32_TFLOPS
            ent["bound"], ent["frac"], ent["peak_tflops"] = "mfma", ent["tflops"] / peak, peak
        else:
            ent["bound"], ent["frac"] = "hbm", ent["gbs"] / PEAK_HBM_GBS
            if "tflops" in ent:                 # executed flops of an HBM-bound MFMA family, for the record
                ent["executed_tflops"] = ent.pop("tflops")
        fams[name] = ent
    return fams


def epoch_bench(args):
    """SURVEY §8(f) row 1 measured: one epoch of the reference's training script (test/ResNet_CIFAR10_cuda.py:173-236) on the
    hip device -- shuffled DataLoader over 50 000 images, augmentation decisions on the host, pinned double-buffered upload
    beside the previous step, crop / flip / erase / clip and label smoothing as kernels, the captured step, the ragged last
    batch (50 000 = 195 x 256 + 80) through a second captured graph, losses read back once -- against the same step on a
    batch that is resident in HBM.  The host stages are also timed by themselves: if the epoch runs below 90 % of the
    resident rate, `bound_stage` names the largest one."""
    from deepflows_amd import backend_api, workloads
    from deepflows_amd.graph_step import CapturedTrainStep
    from deepflows_amd.tensor import Graph
    from deepflows_amd.utils.data import data_loader
    from deepflows_amd.utils.data_pipeline import DeviceBatchPipeline, draw_decisions
    from deepflows_amd.utils.train_loop import EpochTrainer
    batch = args.batch or 256
    n_img = int(args.epoch_images)
    dev = backend_api.Device("hip")
    mod = dev.mod
    model, opt, crit, xt, yt = build_job("resnet18", dev, 0, batch, register_blocks=(args.registration == "all"),
                                         model_kind=args.model)
    rng = np.random.default_rng(99)
    X = np.clip(rng.standard_normal((n_img, 3, 32, 32), dtype=np.float32), -1, 1)
    Y = rng.integers(0, 10, n_img)
    # the resident-batch rate of the same job (what the headline line measures)
    for _ in range(3):
        workloads.train_step(model, crit, opt, xt, yt)
        Graph.free_graph()
    cs = CapturedTrainStep(model, crit, opt, xt, yt)
    for _ in range(3):
        cs.replay()
    mod.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        cs.replay()
    mod.synchronize()
    resident = batch * 100 / (time.perf_counter() - t0)
    cs.destroy()
    trainer = EpochTrainer(model, crit, opt, dev, num_classes=10, eps=0.05, augment=True, capture=True, tail="capture")
    np.random.seed(5)
    loader = data_loader(X, Y, batch, shuffle=True)
    trainer.run_epoch(loader, 0, 30, max_batches=8)                    # captures the full-size step
    tail = n_img % batch
    if tail:
        trainer.run_epoch(data_loader(X[:tail], Y[:tail], batch, shuffle=False), 0, 30)   # and the tail-size one
    mod.synchronize()
    times = []
    for ep in range(4):
        t0 = time.perf_counter()
        losses = trainer.run_epoch(loader, ep + 1, 30)
        mod.synchronize()
        times.append(time.perf_counter() - t0)
    epoch_s = min(times)
    # the host stages by themselves, per full batch
    def per_batch(fn, reps=40):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) * 1e3 / reps
    it = iter(loader)
    # (a short --epoch-images leaves fewer full batches than the default 40 repetitions)
    stage_loader = per_batch(lambda: next(it), reps=max(1, min(40, n_img // batch - 1)))
    stage_decisions = per_batch(lambda: draw_decisions(batch, 32, 32, 1, 30))
    pipe = DeviceBatchPipeline(dev, (batch, 3, 32, 32))
    xb, yb = X[:batch], Y[:batch]

    def submit_deliver():
        pipe.submit(xb, yb, 1, 30)
        pipe.deliver(xt, yt)
    stage_submit = per_batch(submit_deliver) - stage_decisions
    mod.synchronize()
    value = n_img / epoch_s
    stages = {"loader_next_ms": stage_loader, "augment_decisions_ms": stage_decisions, "pack_and_upload_ms": stage_submit,
              "gpu_step_ms": batch / resident * 1e3}
    host = {k: v for k, v in stages.items() if k != "gpu_step_ms"}
    bound = None
    if value < 0.9 * resident:
        k = max(host, key=host.get)
        bound = f"{k} ({host[k]:.3f} ms per batch; the host's stages add up to {sum(host.values()):.3f} ms against {stages['gpu_step_ms']:.3f} ms of GPU work)"
    print(json.dumps({"metric": "epoch images/sec, ResNet-18 CIFAR-10 bs=%d through DataLoader + device input pipeline" % batch,
                      "value": value, "unit": "images/s", "n_gpus": 1, "epoch_images": n_img, "epoch_s": epoch_s,
                      "epochs_timed": len(times), "epoch_s_all": [round(t, 5) for t in times], "batches": len(losses), "tail_batch": tail,
                      "resident_batch_images_per_s": resident, "fraction_of_resident": value / resident,
                      "stages_ms_per_batch": {k: round(v, 4) for k, v in stages.items()}, "bound_stage": bound,
                      "final_loss": float(losses[-1]), "model": args.model, "data": "synthetic", "dtype": "f32",
                      "higher_is_better": True}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="resnet18",
                    help="resnet18 = the BASELINE headline (default); cnn / mlp = BASELINE.json configs 2-3; "
                         "resnet18_wide = test/ResNet.py's 64-512 channel model (stress shape)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's own)")
    ap.add_argument("--dgrad", choices=["overwrite", "accumulate"], default="overwrite",
                    help="conv input-gradient mode: reference-exact (default) or mathematically correct")
    ap.add_argument("--matmul-dtype", choices=["f32", "bf16"], default="f32",
                    help="opt-in (never the headline, outside the 1e-4 parity contract): bf16 operands with fp32 accumulation in "
                         "the register-operand forward convolutions' and the grouped weight gradients' matrix products "
                         "(HipModule.conv_matmul_dtype)")
    ap.add_argument("--registration", choices=["all", "f6"], default="all",
                    help="all = every block registered (62 tensors); f6 = the reference script's 5 tensors")
    ap.add_argument("--model", choices=["plain", "fused"], default="plain",
                    help="resnet18: plain = the model written as the reference's script writes it (default: what an unmodified "
                         "user model gets); fused = the forward() rewritten around the fused entry points")
    ap.add_argument("--epoch", action="store_true",
                    help="time the per-epoch loop of the reference's script instead (test/ResNet_CIFAR10_cuda.py:173-236): 50 000 "
                         "synthetic CIFAR-shaped images through DataLoader -> DeviceBatchPipeline -> captured step (+ ragged tail); "
                         "prints its own JSON line (images/s of the whole epoch, next to the resident-batch rate of the same job)")
    ap.add_argument("--epoch-images", type=int, default=50000)
    ap.add_argument("--no-other-model", action="store_true",
                    help="do not time the other model class beside the headline (profiler passes: every launch of the run is then "
                         "one of the 3 warm-up + 2 + K replayed + K eager steps the counters are divided by)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="do not time the BASELINE's other single-GPU configurations (CNN, MLP, accumulate mode) behind the headline")
    ap.add_argument("--pmc-traffic", default=None,
                    help="PMC summary (tools/pmc_summary.py --json) to quote for traffic / executed flops instead of the "
                         "newest matching file under profiles/")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the step from a hipGraph")
    ap.add_argument("--steps-per-graph", type=int, default=1, choices=(1, 2, 4, 8),
                    help="consecutive steps held by ONE captured graph (graph_step.CapturedTrainStep: a graph launch costs 5-8 us "
                         "of device time whatever it holds); 1 = a graph per step (the headline).  --steps / --warmup should be "
                         "multiples of it; single process only")
    ap.add_argument("--sustain-ms", type=float, default=6500.0,
                    help="after the K timed steps: this many milliseconds of further replays of the captured step in ten "
                         "timed chunks (sustained / median / best ms per step inside `config`); longer than the 5 s interval of "
                         "an outside SMI sampler on purpose; 0 = skip")
    ap.add_argument("--kernels-in-line", action="store_true",
                    help="keep the per-family and per-kernel tables in the JSON line (tools/profile_round.sh); default: they go "
                         "to the side file --kernels-file and the line stays short enough for the driver's record")
    ap.add_argument("--kernels-file", default=None,
                    help="side file of the per-family / per-kernel tables (default gpurun_out/bench_kernels_<workload>.json)")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="untimed replays of the captured step in front of the timed region, in milliseconds of wall time: the "
                         "clock governor needs ~0.2 s of load to reach the clock it then holds (tools/soak.py: 5 000 replays); "
                         "0 = the two warm replays only")
    ap.add_argument("--no-prof-gate", action="store_true",
                    help="per-family event pass with the queue running dry between launches (host-paced)")
    ap.add_argument("--overlap", action="store_true",
                    help="data parallel: two graphs, the all-reduce of layer4 + fc beside the rest of the backward "
                         "pass (costs ~0.1 ms of split / event overhead per step: only pays for slow links)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: this process becomes the spawner of the N ranks (same command line, one child per
        # GPU) and never touches HIP itself -- nothing of the package has been imported yet, and launch.py is
        # standard library only (loaded by path so that not even the package's __init__ runs here)
        import importlib.util
        spec = importlib.util.spec_from_file_location("_df_launch", os.path.join(ROOT, "deepflows_amd", "launch.py"))
        launch = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launch)
        rc, _ = launch.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        sys.exit(rc)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        sys.exit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.epoch:
        if world != 1 or args.workload != "resnet18":
            sys.exit("--epoch times the ResNet-18 script's loop on one GPU")
        epoch_bench(args)
        return

    from deepflows_amd import backend_api, parallel, workloads
    from deepflows_amd.tensor import Graph

    wl = WORKLOADS[args.workload]
    batch = args.batch or wl["batch"]
    dev = backend_api.Device("hip")            # raises when the extension / GPU is missing
    mod = dev.mod
    mod.conv_dgrad_mode = args.dgrad
    mod.conv_matmul_dtype = args.matmul_dtype
    comm = parallel.init_communicator(dev)
    model, opt, crit, xt, yt = build_job(args.workload, dev, rank, batch, register_blocks=(args.registration == "all"),
                                         model_kind=args.model)
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
    # order, one host call per step instead of ~100
    captured = None
    overlapped = False
    settle_replays = 0
    # (under rocprofv3 every replay is traced: the counter passes divide by a fixed number of steps, tools/profile_round.sh)
    profiled_run = args.no_prof_gate
    if not args.no_graph and sync is not None and args.overlap and args.workload == "resnet18":
        # data parallel: two graphs split where layer4's gradients are complete, the all-reduce of
        # that bucket runs on the communication stream beside the rest of the backward pass
        try:
            from deepflows_amd.graph_step import OverlappedCapturedTrainStep
            late, early = workloads.late_and_early_parameters(model, opt.params)
            split_buckets = [parallel.GradientBucket(late, comm), parallel.GradientBucket(early, comm)]
            captured = OverlappedCapturedTrainStep(model, crit, opt, xt, yt, *split_buckets)
            for _ in range(2):
                loss = captured.replay()
            mod.synchronize()
            overlapped = True
        except Exception as e:
            print(f"[bench] overlapped capture failed ({type(e).__name__}: {e}); single-bucket path", file=sys.stderr)
            captured = None
            opt.device_hyper = False
            # (the two buckets took over every parameter's grad_slot: hand the slots back to the single bucket, or its sync()
            # would find every gradient outside its slices and pack all of them on every step)
            for b in locals().get("split_buckets", []):
                b.release()
            bucket.attach()
    if captured is None and not args.no_graph:
        try:
            from deepflows_amd.graph_step import CapturedTrainStep
            spg = args.steps_per_graph if sync is None else 1
            captured = CapturedTrainStep(model, crit, opt, xt, yt, sync=sync, steps_per_graph=spg)
            for _ in range(2 * spg):
                loss = captured.replay()
            mod.synchronize()
            if args.settle_ms > 0 and not profiled_run:
                # (the SAME number of replays on every rank -- a replay may contain the all-reduce: 20 timed ones, then as many
                # as the slowest rank's estimate asks for)
                t_settle = time.perf_counter()
                for _ in range(20):
                    loss = captured.replay()
                mod.synchronize()
                per = max(1e-6, (time.perf_counter() - t_settle) / 20)
                more = int(comm.max_float(float(min(5000, int(args.settle_ms * 1e-3 / per)))))
                more -= more % spg                       # (whole graphs: the timed region starts on a launch)
                for _ in range(more):
                    loss = captured.replay()
                mod.synchronize()
                settle_replays = 20 + more
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

    # sustained leg (SURVEY §8d "report median and best"; VERDICT r4 missing #4): >= --sustain-ms of further replays of the
    # SAME captured step in ten timed chunks, each between synchronize pairs -- two seconds of continuous GPU work that an
    # outside observer (the driver's SMI sampler) can see, and the spread the one K-step region cannot show.  `ms_per_step`
    # / `value` stay the K-step region above.
    sustained = None
    if captured is not None and args.sustain_ms > 0 and not profiled_run:
        per = max(1e-6, wall / args.steps)
        chunk = int(comm.max_float(float(max(args.steps, min(20000, int(args.sustain_ms * 1e-3 / per / 10) + 1)))))
        chunk_ms = []
        for _ in range(10):
            comm.barrier()
            mod.synchronize()
            tc = time.perf_counter()
            for _ in range(chunk):
                captured.replay()
            mod.synchronize()
            chunk_ms.append(comm.max_float(time.perf_counter() - tc) * 1e3 / chunk)
        srt = sorted(chunk_ms)
        sustained = {"replays": 10 * chunk, "chunks": 10, "replays_per_chunk": chunk,
                     "sustained_ms_per_step": sum(chunk_ms) / len(chunk_ms),
                     "median_ms_per_step": 0.5 * (srt[4] + srt[5]), "best_ms_per_step": srt[0],
                     "worst_ms_per_step": srt[-1], "chunk_ms_per_step": [round(c, 5) for c in chunk_ms]}

    # what the exchange costs inside the step (data parallel, or DF_FORCE_COMM=1 DF_FORCE_SYNC=1 on one GPU): the same job
    # captured once more WITHOUT an exchange (a twin model, gradients in arrays of their own) and replayed K times, against K
    # more replays of the step with it -- same process, same box, back to back
    exchange_us = None
    if sync is not None and captured is not None and comm.world == 1:
        try:
            from deepflows_amd.graph_step import CapturedTrainStep

            def replays(cs):
                for _ in range(2):
                    cs.replay()
                mod.synchronize()
                t = time.perf_counter()
                for _ in range(args.steps):
                    cs.replay()
                mod.synchronize()
                return (time.perf_counter() - t) * 1e6 / args.steps
            m2, o2, c2, x2, y2 = build_job(args.workload, dev, rank, batch, register_blocks=(args.registration == "all"),
                                           model_kind=args.model)
            for p2 in o2.params:
                p2.grad_slot = None
            for _ in range(2):
                workloads.train_step(m2, c2, o2, x2, y2)
                Graph.free_graph()
            cap2 = CapturedTrainStep(m2, c2, o2, x2, y2)
            # (alternating, the best of three each: one pair differs by +-10 us from run to run)
            with_us, without_us = float("inf"), float("inf")
            for _ in range(3):
                with_us = min(with_us, replays(captured))
                without_us = min(without_us, replays(cap2))
            exchange_us = with_us - without_us
            cap2.destroy()
            del m2, o2, cap2
        except Exception as e:
            print(f"[bench] the step without the exchange was not timed ({type(e).__name__}: {e})", file=sys.stderr)

    # the OTHER way of writing the model (see build_job), same job, same K replays: reported beside the headline
    other_kind = "fused" if args.model == "plain" else "plain"
    other_value = None
    if args.workload == "resnet18" and comm.world == 1 and sync is None and captured is not None and not args.no_other_model:
        try:
            from deepflows_amd.graph_step import CapturedTrainStep
            m2, o2, c2, x2, y2 = build_job(args.workload, dev, rank, batch, register_blocks=(args.registration == "all"),
                                           model_kind=other_kind)
            for _ in range(2):
                workloads.train_step(m2, c2, o2, x2, y2)
                Graph.free_graph()
            cap2 = CapturedTrainStep(m2, c2, o2, x2, y2)
            for _ in range(2):
                cap2.replay()
            mod.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                cap2.replay()
            mod.synchronize()
            other_value = batch * args.steps / (time.perf_counter() - t2)
            cap2.destroy()
            del m2, o2, cap2
        except Exception as e:
            print(f"[bench] the {other_kind} model was not timed ({type(e).__name__}: {e})", file=sys.stderr)

    # second pass over the same K steps with per-family HIP events (roofline); events cannot be
    # recorded inside a graph replay, so this pass issues the same step eagerly
    if captured is not None:
        opt.device_hyper = False
    mod.prof_reset()
    mod.prof_enable(True)
    for _ in range(args.steps):
        # the stream is held while the host queues the step with its event pairs and released when all of it is
        # queued: the events time launches that run back to back, as in the replayed graph, not the host's gaps
        gated = not args.no_prof_gate
        if gated:
            mod.prof_gate_close(150)
        eager_step()
        if gated:
            mod.prof_gate_open()
    mod.synchronize()
    mod.prof_enable(False)
    prof = mod.prof_read()
    kprof = mod.prof_read_kernels() if hasattr(mod, "prof_read_kernels") else {}
    # the clock the chip holds under a dense fp32-MFMA load (0.2 s of back-to-back MFMAs on every SIMD): the matrix
    # families are priced against the data-sheet peak (2.4 GHz); this says what that peak is worth on this device
    try:
        clock_mhz, probe_tflops = mod.clock_probe(200.0) if rank == 0 else (None, None)
    except Exception as e:
        print(f"[bench] clock probe failed: {e}", file=sys.stderr)
        clock_mhz = probe_tflops = None
    comm.barrier()
    others = None
    if (args.workload == "resnet18" and args.dgrad == "overwrite" and args.matmul_dtype == "f32" and comm.world == 1 and sync is None and captured is not None
            and not profiled_run and not args.no_other_configs and not args.no_other_model):
        try:
            others = other_configurations(args.steps)
        except Exception as e:
            print(f"[bench] the other configurations were not timed ({type(e).__name__}: {e})", file=sys.stderr)

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = batch * comm.world * args.steps / wall
        fams = family_rooflines(prof, args.steps)
        dom = max(fams, key=lambda k: fams[k]["ms_per_step"]) if fams else None
        roofline = None
        if dom:
            d = fams[dom]
            if d["bound"] == "mfma":
                roofline = {"kernel": dom, "bound": "mfma", "achieved": d["tflops"], "peak": d.get("peak_tflops", PEAK_MFMA_F32_TFLOPS),
                            "unit": "TFLOP/s", "frac": d["frac"], "traffic": None}
            else:
                roofline = {"kernel": dom, "bound": "hbm", "achieved": d["gbs"], "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": d["frac"], "traffic": None}
            roofline["avg_launch_us"] = d["avg_us"]
            roofline["avg_launch_us_with_dispatch"] = d["avg_us_with_dispatch"]
            roofline["timing"] = ("avg_launch_us: HIP event pairs bound to each dispatch of the family "
                                  "(hipExtLaunchKernelGGL: the kernel's own start / end stamps, comparable with "
                                  "rocprofv3 --kernel-trace); avg_launch_us_with_dispatch: stream events around the "
                                  "call; eager pass " +
                                  ("queued behind a gate and released per step (back to back)" if not args.no_prof_gate
                                   else "paced by the host"))
            roofline["launches_per_step"] = d["calls_per_step"]
            pmc, pmc_src = pmc_file(args.workload, args.dgrad, args.pmc_traffic)
            pfam = (pmc or {}).get("families", {}).get(dom)
            roofline["traffic_source"] = pmc_src if pfam or pmc is None else f"{pmc_src}: no entry for {dom}"
            if pfam:
                # counters of a separate rocprofv3 pass of this command on this build: HBM bytes per launch, the flops
                # the matrix cores EXECUTED (padding taps that are skipped are algorithmic but not executed) against the
                # peak over this run's own launch durations, and the matrix-pipe busy fraction inside the profiled pass
                roofline["traffic"] = pfam["hbm_bytes_per_step"] / d["calls_per_step"]
                if d["bound"] == "mfma" and pfam.get("executed_flops_per_step"):
                    ex = pfam["executed_flops_per_step"] / d["calls_per_step"]
                    roofline["executed_flops_per_launch"] = ex
                    roofline["frac_executed"] = ex / (d["avg_us"] * 1e-6) / 1e12 / PEAK_MFMA_F32_TFLOPS
                    roofline["mfma_busy"] = pfam.get("mfma_busy")
            if args.matmul_dtype == "bf16" and not dom.endswith("_bf16"):
                roofline["note"] = ("opt-in bf16 mode: this family keeps its fp32 name and peak although the layer1-2 launches of "
                                    "it run v_mfma_f32_16x16x16_bf16")
            roofline["mfma_clock_mhz_under_load"] = clock_mhz
            roofline["mfma_probe_tflops"] = probe_tflops
            roofline["algorithmic_bytes_per_launch"] = prof[dom]["bytes"] / prof[dom]["calls"]
            roofline["algorithmic_flops_per_launch"] = prof[dom]["flops"] / prof[dom]["calls"]
        exchanged = bucket.total if sync is not None else 0
        rccl_ranks = mod.comm_count()[0] if comm.exchanges else None
        # the whole step priced two ways: the algorithmic contraction flops (BASELINE.md §3 / SURVEY §8d: input gradients
        # counted at the forward's flop count) and the flops the matrix cores EXECUTE (PMC pass of the same kernel sources;
        # the overwrite input gradient keeps 1/9 ... 4/9 of the taps, padding taps are not multiplied)
        # (SURVEY §8d / BASELINE.md §3 figures at the workloads' own batch; the event pass's per-family flops book the overwrite
        # input gradient at its executed count, so their sum is not this figure)
        alg_flops = {"resnet18": 54.06e9, "cnn": 29.93e9, "mlp": 0.1047e9, "resnet18_wide": 852.2e9}[args.workload] * batch / wl["batch"]
        pmc_all, _ = pmc_file(args.workload, args.dgrad, args.pmc_traffic)
        exec_flops = (sum(f.get("executed_flops_per_step", 0.0) for f in pmc_all["families"].values()) if pmc_all else None)
        fam_table = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in
                     sorted(fams.items(), key=lambda kv: -kv[1]["ms_per_step"])}
        # per kernel instantiation (names as rocprofv3 prints them): launches per step, the kernel's own duration,
        # the algorithmic work of the family calls it was the first launch of
        kernel_table = [{"kernel": name, "family": r["family"], "launches_per_step": round(r["launches"] / args.steps, 3),
                         "avg_us": round(r["kernel_ms"] / r["launches"] * 1e3, 3),
                         "alg_gflop_per_launch": round(r["flops"] / r["launches"] / 1e9, 4),
                         "alg_mb_per_launch": round(r["bytes"] / r["launches"] / 1e6, 3)}
                        for name, r in sorted(kprof.items(), key=lambda kv: -kv[1]["kernel_ms"]) if r["launches"]]
        kernels_us = sum(r["kernel_ms"] for r in kprof.values()) * 1e3 / args.steps if kprof else None
        r4 = lambda v: None if v is None else round(v, 4)
        out = {
            "metric": wl["metric"],
            "value": value, "unit": "samples/s" if args.workload == "mlp" else "images/s", "n_gpus": comm.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.matmul_dtype == "f32" else "bf16 operands / f32 accumulation in the layer1-2 forward convolutions and the grouped weight gradients, f32 elsewhere", "data": "synthetic",
            "config": {"workload": wl["text"],
                       "global_batch": batch * comm.world, "batch_per_gpu": batch, "parallelism": f"dp{comm.world}",
                       "settle_replays_untimed": settle_replays, "conv_dgrad": args.dgrad, "conv_matmul_dtype": args.matmul_dtype, "dispatch": ("hipgraph+overlapped-allreduce" if overlapped else "hipgraph") if captured is not None else "eager", "steps_per_graph": getattr(captured, "S", 1) if captured is not None else None, "registered_params": args.registration,
                       "model": args.model if args.workload == "resnet18" else None,
                       "launches_per_step": round(sum(r["launches"] for r in kprof.values()) / args.steps, 2) if kprof else None,
                       "kernel_us_per_step": r4(kernels_us),
                       # ten timed chunks of further replays of the same captured step (--sustain-ms in all): mean, median, best
                       "sustained_ms_per_step": r4(sustained["sustained_ms_per_step"]) if sustained else None,
                       "median_ms_per_step": r4(sustained["median_ms_per_step"]) if sustained else None,
                       "best_ms_per_step": r4(sustained["best_ms_per_step"]) if sustained else None,
                       "worst_ms_per_step": r4(sustained["worst_ms_per_step"]) if sustained else None,
                       "sustained_replays": sustained["replays"] if sustained else None,
                       "sustained_images_per_s": round(batch * comm.world / (sustained["sustained_ms_per_step"] * 1e-3), 1) if sustained else None,
                       # the step's contractions: algorithmic flops and the flops the matrix cores execute (PMC), both as a
                       # fraction of the fp32 matrix peak over ms_per_step
                       "algorithmic_gflop_per_step": r4(alg_flops / 1e9),
                       "executed_gflop_per_step": r4(exec_flops / 1e9) if exec_flops else None,
                       "whole_step_frac_algorithmic": r4(alg_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS),
                       "whole_step_frac_executed": r4(exec_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS) if exec_flops else None,
                       f"{other_kind}_model_images_per_s": r4(other_value),
                       # BASELINE.json configs 2-3 and the accumulate mode, each timed by a child process of this run
                       # (never the headline)
                       "other_configurations": others,
                       "allreduce_elems": exchanged,
                       # the exchange step: inside the captured graph or enqueued behind it; gradients the last sync() had to
                       # copy into the bucket (0: every backward kernel wrote its gradient in place)
                       "exchange_captured": getattr(captured, "exchange_captured", None) if sync is not None else None,
                       "exchange_fallback_reason": getattr(captured, "exchange_fallback_reason", None) if sync is not None else None,
                       "grads_copied_into_bucket": bucket.packed_last if sync is not None else None,
                       # (world 1 only: replayed step with the exchange minus replayed step without, microseconds)
                       "exchange_us_in_graph": exchange_us,
                       # what the communicator itself reports (ncclCommCount), not the launcher's environment
                       "rccl_ranks": rccl_ranks,
                       "collective_library": mod.comm_rccl_path() if sync is not None else None},
            "gpu_ms_per_step": gpu_ms / args.steps, "host_issue_ms_per_step": 1e3 * host_issue / args.steps,
            "final_loss": final_loss,
            "roofline": roofline,
        }
        side = {"kernels": fam_table, "kernel_table": kernel_table, "sustained": sustained}
        if args.kernels_in_line:
            out.update(side)
        else:
            path = args.kernels_file or os.path.join(ROOT, "gpurun_out", f"bench_kernels_{args.workload}"
                                                     f"{'_accumulate' if args.dgrad == 'accumulate' else ''}.json")
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "w") as f:
                    json.dump(dict(side, metric=wl["metric"], ms_per_step=ms_per_step), f, indent=1)
                out["kernels_file"] = os.path.relpath(path, ROOT)
            except OSError as e:
                out["kernels_file"] = f"not written: {e}"
        if comm.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, batch)
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
