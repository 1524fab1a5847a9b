"""GPU: the RCCL path of the data-parallel step with a one-rank communicator (the only world
size a 1-GPU box allows): librccl is dlopen'ed, ncclCommInitRank / ncclAllReduce run on the
library stream, the bucket views feed the fused Adam, and the result equals the no-communicator
step bit for bit (sum over one rank, grad_scale = 1).  Also the captured step with a gradient
exchange (round 4: all-reduce and Adam inside the graph, the gradients formed in the bucket by the backward kernels; the
rounds 1-3 form -- graph up to backward, then the exchange eagerly -- as the fall-back)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_comm(request):
    from deepflows_amd import backend_api, parallel
    dev = backend_api.Device("hip")
    saved = {k: os.environ.get(k) for k in ("DF_FORCE_COMM", "MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    os.environ.update(DF_FORCE_COMM="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + (os.getpid() % 1000)),
                      RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    comm = parallel.init_communicator(dev)
    assert isinstance(comm, parallel.RcclCommunicator), type(comm)
    yield dev, comm
    comm.close()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _cnn_two_steps(dev, comm):
    from deepflows_amd import nn, optim, parallel, workloads
    from deepflows_amd.tensor import Graph, Tensor
    np.random.seed(0)
    model = workloads.CIFAR10_CNN(10, device=dev)
    model.drop.p = 0.0
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
    bucket = parallel.GradientBucket(opt.params, comm)
    opt.grad_scale = 1.0 / comm.world
    rng = np.random.default_rng(3)
    x = rng.standard_normal((8, 3, 32, 32)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 8)]
    use = isinstance(comm, parallel.RcclCommunicator)
    for step in range(2):
        np.random.seed(100 + step)
        loss, _ = workloads.train_step(model, nn.CrossEntropyLoss(), opt, Tensor(x, device=dev), Tensor(y, device=dev),
                                       sync=bucket.sync if use else None)
        Graph.free_graph()
    return float(loss.numpy()[0]), [p.data.numpy().copy() for p in model.parameters()]


def test_rccl_one_rank_matches_local(rccl_comm):
    from deepflows_amd import parallel
    dev, comm = rccl_comm
    la, pa = _cnn_two_steps(dev, comm)
    lb, pb = _cnn_two_steps(dev, parallel.LocalCommunicator())
    assert la == lb
    for a, b in zip(pa, pb):
        assert np.array_equal(a, b)


def test_captured_step_with_gradient_exchange(rccl_comm):
    """two replays must continue exactly like two more eager steps of a twin model"""
    from deepflows_amd import nn, optim, parallel, workloads
    from deepflows_amd.graph_step import CapturedTrainStep
    from deepflows_amd.tensor import Graph, Tensor
    dev, comm = rccl_comm
    rng = np.random.default_rng(3)
    xs = Tensor(rng.standard_normal((8, 3, 32, 32)).astype(np.float32), device=dev)
    ys = Tensor(np.eye(10, dtype=np.float32)[rng.integers(0, 10, 8)], device=dev)
    crit = nn.CrossEntropyLoss()

    def twin():
        np.random.seed(1)
        m = workloads.ResNet18(device=dev, register_blocks=True)
        o = optim.Adam(workloads.all_block_parameters(m), lr=1e-3)
        return m, o, parallel.GradientBucket(o.params, comm)

    m1, o1, b1 = twin()
    for _ in range(3):
        workloads.train_step(m1, crit, o1, xs, ys, sync=b1.sync)
        Graph.free_graph()
    m2, o2, b2 = twin()
    workloads.train_step(m2, crit, o2, xs, ys, sync=b2.sync)
    Graph.free_graph()
    assert b2.packed_last == 0, "the backward kernels did not write every gradient into the bucket"
    cs = CapturedTrainStep(m2, crit, o2, xs, ys, sync=b2.sync)
    # RCCL lets a stream capture record ncclAllReduce: the step is ONE graph (backward, all-reduce, Adam)
    assert cs.exchange_captured, cs.exchange_fallback_reason
    cs.replay()
    cs.replay()
    dev.mod.comm_check()                            # (ncclCommGetAsyncError: the communicator is healthy)
    assert o1.t == o2.t == 4
    for a, b in zip(o1.params, o2.params):
        assert np.array_equal(a.data.numpy(), b.data.numpy())
    # and the same with the exchange enqueued behind the graph (what an uncapturable collective falls back to)
    m3, o3, b3 = twin()
    workloads.train_step(m3, crit, o3, xs, ys, sync=b3.sync)
    Graph.free_graph()
    cs3 = CapturedTrainStep(m3, crit, o3, xs, ys, sync=b3.sync, capture_exchange=False)
    assert not cs3.exchange_captured
    cs3.replay()
    cs3.replay()
    for a, b in zip(o1.params, o3.params):
        assert np.array_equal(a.data.numpy(), b.data.numpy())


def test_gradients_are_formed_in_the_bucket(rccl_comm):
    """every parameter gradient of a ResNet-18 / CNN step is written by its backward kernel into the parameter's slice of
    the flat bucket (no pack launch); a parameter that collects two contributions is copied in"""
    from deepflows_amd import nn, optim, parallel, workloads
    from deepflows_amd.tensor import Graph, Tensor
    dev, comm = rccl_comm
    rng = np.random.default_rng(4)
    for make, shape in ((lambda: workloads.ResNet18Plain(device=dev, register_blocks=True), (32, 3, 32, 32)),
                        (lambda: workloads.CIFAR10_CNN(10, device=dev), (16, 3, 32, 32)),
                        (lambda: workloads.MLP((784, 256, 10), device=dev), (16, 784))):
        np.random.seed(2)
        m = make()
        o = optim.Adam(workloads.all_block_parameters(m) if hasattr(m, "layer1") else m.parameters(), lr=1e-3)
        b = parallel.GradientBucket(o.params, comm)
        xs = Tensor(rng.standard_normal(shape).astype(np.float32), device=dev)
        ys = Tensor(np.eye(10, dtype=np.float32)[rng.integers(0, 10, shape[0])], device=dev)
        dev.mod.prof_reset()
        dev.mod.prof_enable(True)
        workloads.train_step(m, nn.CrossEntropyLoss(), o, xs, ys, sync=b.sync)
        dev.mod.synchronize()
        dev.mod.prof_enable(False)
        Graph.free_graph()
        assert b.packed_last == 0, f"{type(m).__name__}: {b.packed_last} gradients were copied into the bucket"
        assert not dev.mod.prof_read().get("bucket_pack", {}).get("calls"), "a pack launch ran"


def test_overlapped_step_matches_eager_exchange(rccl_comm):
    """OverlappedCapturedTrainStep (two graphs, all-reduce of the late bucket on the communication
    stream) continues exactly like eager steps with a single bucket."""
    from deepflows_amd import nn, optim, parallel, workloads
    from deepflows_amd.graph_step import OverlappedCapturedTrainStep
    from deepflows_amd.tensor import Graph, Tensor
    dev, comm = rccl_comm
    rng = np.random.default_rng(3)
    xs = Tensor(rng.standard_normal((32, 3, 32, 32)).astype(np.float32), device=dev)
    ys = Tensor(np.eye(10, dtype=np.float32)[rng.integers(0, 10, 32)], device=dev)
    crit = nn.CrossEntropyLoss()

    def twin():
        np.random.seed(1)
        m = workloads.ResNet18(device=dev, register_blocks=True)
        return m, optim.Adam(workloads.all_block_parameters(m), lr=1e-3)

    m1, o1 = twin()
    b1 = parallel.GradientBucket(o1.params, comm)
    for _ in range(4):
        workloads.train_step(m1, crit, o1, xs, ys, sync=b1.sync)
        Graph.free_graph()
    m2, o2 = twin()
    b2 = parallel.GradientBucket(o2.params, comm)
    workloads.train_step(m2, crit, o2, xs, ys, sync=b2.sync)
    Graph.free_graph()
    late, early = workloads.late_and_early_parameters(m2, o2.params)
    assert len(late) > 0 and len(early) > 0 and len(late) + len(early) == len(o2.params)
    assert sum(int(p.size) for p in late) > 2 * sum(int(p.size) for p in early)
    step = OverlappedCapturedTrainStep(m2, crit, o2, xs, ys, parallel.GradientBucket(late, comm),
                                       parallel.GradientBucket(early, comm))
    for _ in range(3):
        step.replay()
    assert o1.t == o2.t == 5
    # (bit for bit as long as both half sweeps are planned like the single one: a group of less than a quarter of
    # the chip's workgroups would be re-planned as stand-alone launches with other pixel slabs)
    for a, b in zip(o1.params, o2.params):
        assert np.array_equal(a.data.numpy(), b.data.numpy())


def test_bench_exchange_path_runs_on_one_gpu():
    """bench.py exactly as the driver runs it on N > 1 GPUs, except that the world is one rank: RcclCommunicator
    (rendezvous skipped at world 1, ncclCommInitRank, barrier and max-over-ranks as one-element collectives), the
    captured step ending after backward, pack + ncclAllReduce + Adam behind it.  All 2 797 610 parameters go through
    the bucket, the collective library is ROCm's own, and nothing imported torch."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", DF_FORCE_COMM="1", DF_FORCE_SYNC="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 1000))
    code = ("import sys, runpy; sys.argv = ['bench.py', '--steps', '3', '--warmup', '2', '--no-cpu-baseline', '--kernels-in-line', '--sustain-ms', '0']; "
            "runpy.run_path(%r, run_name='__main__'); assert 'torch' not in sys.modules, 'torch was imported'"
            % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["config"]["allreduce_elems"] == 2797610
    assert d["config"]["dispatch"] == "hipgraph" and d["n_gpus"] == 1
    assert d["config"]["collective_library"].startswith("/opt/rocm"), d["config"]["collective_library"]
    assert d["config"]["rccl_ranks"] == 1                 # ncclCommCount of the communicator itself
    assert "rccl_allreduce" in d["kernels"]
    assert d["value"] > 0 and np.isfinite(d["final_loss"])
    # round 4: the exchange is part of the captured graph, the gradients are formed in the bucket, its time is reported
    assert d["config"]["exchange_captured"] is True, d["config"]["exchange_fallback_reason"]
    assert d["config"]["grads_copied_into_bucket"] == 0
    # (a timing: measured -12 ... +6 us over six boxes, best of three alternating pairs; round 3's pack launch + eager
    # exchange behind the graph was ~40 us -- the bound only has to tell the two apart)
    assert d["config"]["exchange_us_in_graph"] is not None and d["config"]["exchange_us_in_graph"] < 35.0
    print("exchange at world 1 inside the captured step: %+.1f us per step" % d["config"]["exchange_us_in_graph"])
