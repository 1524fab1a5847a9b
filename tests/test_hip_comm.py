"""GPU: the RCCL path of the data-parallel step with a one-rank communicator (the only world
size a 1-GPU box allows): librccl is dlopen'ed, ncclCommInitRank / ncclAllReduce run on the
library stream, the bucket views feed the fused Adam, and the result equals the no-communicator
step bit for bit (sum over one rank, grad_scale = 1)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["DF_ROOT"])
from deepflows_amd import backend_api, nn, optim, parallel, workloads
from deepflows_amd.tensor import Graph, Tensor
dev = backend_api.Device("hip")
use_comm = os.environ.get("DF_FORCE_COMM") == "1"
comm = parallel.init_communicator(dev)
assert isinstance(comm, parallel.RcclCommunicator) == use_comm, type(comm)
np.random.seed(0)
model = workloads.CIFAR10_CNN(10, device=dev)
model.drop.p = 0.0
opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
bucket = parallel.GradientBucket(opt.params, comm)
opt.grad_scale = 1.0 / comm.world
rng = np.random.default_rng(3)
x = rng.standard_normal((8, 3, 32, 32)).astype(np.float32)
y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 8)]
for step in range(2):
    np.random.seed(100 + step)
    loss, _ = workloads.train_step(model, nn.CrossEntropyLoss(), opt, Tensor(x, device=dev), Tensor(y, device=dev),
                                   sync=bucket.sync if use_comm else None)
    Graph.free_graph()
out = [p.data.numpy().tolist() for p in model.parameters()]
json.dump({"loss": float(loss.numpy()[0]), "params": out}, open(os.environ["DF_OUT"], "w"))
comm.close()
'''


def _run(tmp_path, force):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    out = tmp_path / f"out{force}.json"
    env = dict(os.environ, DF_ROOT=ROOT, DF_OUT=str(out), DF_FORCE_COMM=str(force), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29600 + (os.getpid() % 1000)), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    return json.load(open(out))


def test_rccl_one_rank_matches_local(tmp_path):
    a = _run(tmp_path, 1)
    b = _run(tmp_path, 0)
    assert a["loss"] == b["loss"]
    for pa, pb in zip(a["params"], b["params"]):
        assert np.array_equal(np.array(pa), np.array(pb))
