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


def test_dill_written_checkpoint_loads_without_dill(cpu_oracle, monkeypatch):
    """The reference saves through dill when it is installed (utils/model_utils.py:10-16; requirements.txt lists
    it): tests/golden/ck_reference_mlp_dill.pkl is its file byte for byte (dill._dill._create_array records).  The
    restricted loader reads it through shims and never imports dill."""
    import sys
    from deepflows_amd.utils import model_utils
    monkeypatch.setitem(sys.modules, "dill", None)            # `import dill` would raise from here on
    monkeypatch.setitem(sys.modules, "dill._dill", None)
    raw = open(os.path.join(GOLDEN, "ck_reference_mlp_dill.pkl"), "rb").read()
    assert b"dill._dill" in raw and b"_create_array" in raw
    ck = model_utils.read_checkpoint(os.path.join(GOLDEN, "ck_reference_mlp_dill.pkl"))
    exp = np.load(os.path.join(GOLDEN, "ck_reference_mlp_expect.npz"))
    for name, a in ck["model_parameters"].items():
        assert isinstance(a, np.ndarray) and np.array_equal(a, exp["param." + name]), name
    assert ck["optimizer_state"]["t"] == int(exp["t"]) and ck["epoch"] == 3
    for i, (v, s) in enumerate(zip(ck["optimizer_state"]["v"], ck["optimizer_state"]["s"])):
        assert np.array_equal(v, exp[f"v.{i}"]) and np.array_equal(s, exp[f"s.{i}"])


def test_shipped_reference_checkpoint_loads_directly():
    """test/checkpoints-cifar10cuda_70%/cifar10_cnn_cuda_checkpoint.pkl of the reference (written by its CUDA run
    through dill): loaded as it lies, compared with the per-parameter shapes and sums the generator recorded.
    Only where the reference tree exists (the build container)."""
    path = "/root/reference/test/checkpoints-cifar10cuda_70%/cifar10_cnn_cuda_checkpoint.pkl"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    from deepflows_amd.utils import model_utils
    summ = json.load(open(os.path.join(GOLDEN, "ck_shipped_cnn_summary.json")))
    ck = model_utils.read_checkpoint(path)
    assert ck["epoch"] == summ["epoch"] and set(ck["model_parameters"]) == set(summ["parameters"])
    for k, meta in summ["parameters"].items():
        a = ck["model_parameters"][k]
        assert list(a.shape) == meta["shape"]
        assert abs(float(np.asarray(a, np.float64).sum()) - meta["sum"]) <= 1e-9 * max(1.0, abs(meta["sum"]))


def test_load_checkpoint_keeps_the_device_buffers(cpu_oracle, tmp_path):
    """A load overwrites parameters, Adam moments and running statistics IN PLACE (same device handles): a captured
    step (graph_step.CapturedTrainStep) has their raw pointers baked into the graph."""
    from deepflows_amd import nn, optim
    from deepflows_amd.utils import load_checkpoint, save_checkpoint
    np.random.seed(3)
    src = _mlp(nn, cpu_oracle)
    sopt = optim.Adam(src.parameters(), lr=1e-3)
    path = str(tmp_path / "ck.pkl")
    save_checkpoint(src, sopt, save_path=path)
    np.random.seed(4)
    dst = _mlp(nn, cpu_oracle)
    dopt = optim.Adam(dst.parameters(), lr=1.0)
    handles = [id(p.data._handle) for _, p in dst.named_parameters()] + [id(v._handle) for v in dopt.v] + \
              [id(s._handle) for s in dopt.s]
    load_checkpoint(dst, dopt, path)
    after = [id(p.data._handle) for _, p in dst.named_parameters()] + [id(v._handle) for v in dopt.v] + \
            [id(s._handle) for s in dopt.s]
    assert handles == after
    for (_, a), (_, b) in zip(src.named_parameters(), dst.named_parameters()):
        assert np.array_equal(a.numpy(), b.numpy())
