"""Checkpoint format compatibility: the fixtures were written by the reference's
own save_checkpoint (oracle/gen_checkpoint_golden.py, run where the reference is available)."""
import json
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN, assert_close


def _mlp(nn, dev):
    class MLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(12, 8, device=dev)
            self.fc2 = nn.Linear(8, 3, device=dev)

        def forward(self, x):
            return self.fc2(nn.functional.relu(self.fc1(x)))
    return MLP()


def _resume_and_check(dev):
    from deepflows_amd import nn, optim
    from deepflows_amd.tensor import Graph, Tensor
    from deepflows_amd.utils import load_checkpoint
    exp = np.load(os.path.join(GOLDEN, "ck_reference_mlp_expect.npz"))
    np.random.seed(123)                                  # different init: everything must come from the file
    model = _mlp(nn, dev)
    opt = optim.Adam(model.parameters(), lr=0.5)
    info = load_checkpoint(model, opt, os.path.join(GOLDEN, "ck_reference_mlp.pkl"))
    assert info["epoch"] == 3 and info["loss"] is not None
    assert opt.lr == 1e-2 and opt.weight_decay == 5e-4 and opt.t == int(exp["t"])
    for name, p in model.named_parameters():
        assert np.array_equal(p.numpy(), exp["param." + name]), name
    for i in range(len(opt.v)):
        assert np.array_equal(opt.v[i].numpy(), exp[f"v.{i}"]) and np.array_equal(opt.s[i].numpy(), exp[f"s.{i}"])
    # one resumed training step reproduces the reference's next step
    out = model(Tensor(exp["x"], device=dev))
    loss = nn.CrossEntropyLoss()(out, Tensor(exp["y"], device=dev))
    opt.zero_grad()
    loss.backward()
    opt.step()
    Graph.free_graph()
    assert abs(float(loss.numpy()[0]) - float(exp["next_loss"])) <= 1e-5 * max(1.0, abs(float(exp["next_loss"])))
    for name, p in model.named_parameters():
        assert_close(p.numpy(), exp["next." + name], 1e-5, name)


def test_reference_checkpoint_resumes_on_the_oracle_device(cpu_oracle):
    _resume_and_check(cpu_oracle)


@pytest.mark.gpu
def test_reference_checkpoint_resumes_on_hip(hip_device):
    _resume_and_check(hip_device)


def test_our_checkpoint_has_the_reference_schema_and_loads_there(cpu_oracle, tmp_path):
    from deepflows_amd import nn, optim
    from deepflows_amd.utils import load_checkpoint, save_checkpoint
    np.random.seed(1)
    model = _mlp(nn, cpu_oracle)
    opt = optim.Adam(model.parameters(), lr=3e-3)
    path = str(tmp_path / "ck.pkl")
    save_checkpoint(model, opt, epoch=9, loss=1.25, save_path=path)
    ck = pickle.load(open(path, "rb"))                   # plain pickle, as the reference (or dill) reads it
    assert set(ck) >= {"epoch", "loss", "model_parameters", "optimizer_state"}
    assert set(ck["model_parameters"]) == {n for n, _ in model.named_parameters()}
    assert all(isinstance(a, np.ndarray) and a.dtype == np.float32 for a in ck["model_parameters"].values())
    st = ck["optimizer_state"]
    assert st["type"] == "Adam" and st["t"] == 1 and len(st["v"]) == len(st["s"]) == 4
    # the reference's own load_checkpoint accepted a file written by this function when the fixtures
    # were generated (it cannot be imported here)
    rt = json.load(open(os.path.join(GOLDEN, "ck_roundtrip.json")))
    assert rt["reference_loads_our_checkpoint"] and rt["epoch"] == 9 and rt["lr_restored"] == 3e-3
    # and it round-trips through this package
    np.random.seed(2)
    other = _mlp(nn, cpu_oracle)
    load_checkpoint(other, None, path)
    for (_, a), (_, b) in zip(model.named_parameters(), other.named_parameters()):
        assert np.array_equal(a.numpy(), b.numpy())


def test_loader_refuses_code_in_a_checkpoint(cpu_oracle, tmp_path):
    from deepflows_amd import nn
    from deepflows_amd.utils import load_checkpoint

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    path = str(tmp_path / "evil.pkl")
    pickle.dump({"model_parameters": {"fc1.weight": Evil()}, "epoch": 0, "loss": None}, open(path, "wb"))
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint(_mlp(nn, cpu_oracle), None, path)
