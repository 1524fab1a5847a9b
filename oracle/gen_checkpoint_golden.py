"""TEST INFRASTRUCTURE — checkpoint fixtures.  Runs ONLY in the build container (reference at
/root/reference).  (1) Builds the reference's own MLP + Adam on its cpu device, takes two
training steps and writes a checkpoint with the REFERENCE's save_checkpoint
(DeepFlows/utils/model_utils.py:19-83) -> tests/golden/ck_reference_mlp.pkl (a pickled dict of
NumPy arrays: data, not source), plus the parameters it should restore ->
tests/golden/ck_reference_mlp_expect.npz.  (2) Writes a checkpoint with THIS repository's
save_checkpoint (oracle NumPy device) and loads it with the REFERENCE's load_checkpoint into a
fresh reference model; the comparison result is recorded in tests/golden/ck_roundtrip.json.

    python oracle/gen_checkpoint_golden.py
"""
import json
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)


def reference_part():
    import DeepFlows
    from DeepFlows import nn, optim, backend_api
    from DeepFlows.tensor import Tensor, Graph
    from DeepFlows.utils import model_utils as ref_mu

    dev = backend_api.Device("cpu")

    class MLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(12, 8, device="cpu")
            self.fc2 = nn.Linear(8, 3, device="cpu")

        def forward(self, x):
            return self.fc2(nn.functional.relu(self.fc1(x)))

    np.random.seed(42)
    model = MLP()
    opt = optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-4)
    crit = nn.CrossEntropyLoss()
    rng = np.random.default_rng(7)
    x = rng.standard_normal((6, 12)).astype(np.float32)
    y = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 6)]
    for _ in range(2):
        out = model(Tensor(x, device=dev))
        loss = crit(out, Tensor(y, device=dev))
        opt.zero_grad()
        loss.backward()
        opt.step()
        Graph.free_graph()
    path = os.path.join(OUT, "ck_reference_mlp.pkl")
    ref_mu.save_checkpoint(model, opt, epoch=3, loss=float(loss.data.numpy()[0]), save_path=path)
    # the file exactly as the reference wrote it (through dill when it is installed, as its requirements.txt
    # asks): kept byte for byte as a second fixture, the loader must read it without importing dill
    import shutil
    shutil.copyfile(path, os.path.join(OUT, "ck_reference_mlp_dill.pkl"))
    ck = pickle.load(open(path, "rb"))
    # re-write with the standard pickler so that the fixture does not depend on dill at all
    pickle.dump(ck, open(path, "wb"), protocol=4)
    exp = {"param." + k: v for k, v in ck["model_parameters"].items()}
    for i, (v, s) in enumerate(zip(ck["optimizer_state"]["v"], ck["optimizer_state"]["s"])):
        exp[f"v.{i}"], exp[f"s.{i}"] = v, s
    exp["t"] = np.array(ck["optimizer_state"]["t"])
    exp["x"], exp["y"] = x, y
    # one more reference step from the restored state: what a resumed run must produce
    out = model(Tensor(x, device=dev))
    loss = crit(out, Tensor(y, device=dev))
    opt.zero_grad()
    loss.backward()
    opt.step()
    Graph.free_graph()
    exp["next_loss"] = np.array(float(loss.data.numpy()[0]), dtype=np.float64)
    for n, p in model.named_parameters():
        exp["next." + n] = p.data.numpy()
    np.savez(os.path.join(OUT, "ck_reference_mlp_expect.npz"), **exp)
    # the checkpoint the reference ships (test/checkpoints-cifar10cuda_70%/, written by its CUDA run through
    # dill): per-entry shapes and float64 sums, so that a test can check a direct load of that file where it exists
    shipped = "/root/reference/test/checkpoints-cifar10cuda_70%/cifar10_cnn_cuda_checkpoint.pkl"
    if os.path.exists(shipped):
        import dill
        sk = dill.load(open(shipped, "rb"))
        summ = {"epoch": sk.get("epoch"), "loss": sk.get("loss"), "optimizer_type": sk.get("optimizer_state", {}).get("type"),
                "parameters": {k: {"shape": list(v.shape), "sum": float(np.asarray(v, np.float64).sum())}
                               for k, v in sk["model_parameters"].items()}}
        json.dump(summ, open(os.path.join(OUT, "ck_shipped_cnn_summary.json"), "w"), indent=1)
    return MLP, ref_mu, dev


def roundtrip_part(RefMLP, ref_mu, ref_dev):
    # this repository's model on the oracle NumPy device, saved with its own save_checkpoint
    import deepflows_amd as df
    from deepflows_amd import backend_api as bapi, nn, optim
    from deepflows_amd.utils import save_checkpoint
    from oracle.numpy_device import NumpyDevice
    bapi.register_device("cpu", NumpyDevice())
    bapi.set_default_device("cpu")

    class MLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(12, 8, device="cpu")
            self.fc2 = nn.Linear(8, 3, device="cpu")

        def forward(self, x):
            return self.fc2(nn.functional.relu(self.fc1(x)))

    np.random.seed(5)
    mine = MLP()
    opt = optim.Adam(mine.parameters(), lr=3e-3)
    tmp = os.path.join(OUT, "_tmp_ck.pkl")
    save_checkpoint(mine, opt, epoch=9, loss=1.25, save_path=tmp)
    np.random.seed(6)
    ref = RefMLP()
    from DeepFlows import optim as roptim
    ropt = roptim.Adam(ref.parameters(), lr=1.0)
    info = ref_mu.load_checkpoint(ref, ropt, tmp)
    ok = all(np.array_equal(p.data.numpy(), q.data.numpy())
             for (_, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()))
    res = {"reference_loads_our_checkpoint": bool(ok), "epoch": info["epoch"], "loss": info["loss"],
           "lr_restored": ropt.lr, "t_restored": int(ropt.t)}
    os.remove(tmp)
    json.dump(res, open(os.path.join(OUT, "ck_roundtrip.json"), "w"), indent=1)
    print(res)


if __name__ == "__main__":
    cls, mu, dev = reference_part()
    roundtrip_part(cls, mu, dev)
