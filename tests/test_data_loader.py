"""the loader and the epoch loop.

CPU: `deepflows_amd.utils.data` against batch-index sequences the reference's own DataLoader produced under fixed
seeds (tests/golden/g12_epoch.npz, oracle/gen_epoch_golden.py), and the reference's epoch loop
(test/ResNet_CIFAR10_cuda.py:173-236: shuffled loader, augment_batch, label smoothing, Adam, WarmupCosineLR) restated
test-side on the oracle device against the losses the reference itself printed for that loop.
GPU: `utils.train_loop.EpochTrainer` (device pipeline + captured step + ragged tail) against the same reference losses,
and -- at the benchmark's batch size -- bit for bit against the plain eager loop over the same batches."""
import numpy as np
import pytest

from conftest import load_golden, rel_err


def _loader_case(g, ci):
    from deepflows_amd.utils.data import DataLoader, Dataset
    n, bs, shuffle, drop_last, prefetch, seed = (int(v) for v in g[f"order{ci}_meta"])
    X = np.arange(n, dtype=np.float32).reshape(n, 1) * 10
    y = np.arange(n)

    class DS(Dataset):
        def __getitem__(self, index):
            return X[index], y[index]

        def __len__(self):
            return n
    return DataLoader(DS(), bs, bool(shuffle), bool(drop_last), prefetch_size=prefetch), seed, X


def test_batch_order_equals_the_references():
    g = load_golden("g12_epoch.npz")
    for ci in range(int(g["order_cases"])):
        loader, seed, X = _loader_case(g, ci)
        np.random.seed(seed)
        sizes, idx, between = [], [], []
        for _pass in range(2):                       # a RandomSampler draws a new permutation on every pass
            it = iter(loader)
            between.append(np.random.randint(0, 1 << 30))      # pins WHEN the permutation is drawn (prefetch or not)
            for xb, yb in it:
                assert xb.flags["C_CONTIGUOUS"] and np.array_equal(xb, X[yb])
                sizes.append(len(yb))
                idx.append(np.asarray(yb))
        assert sizes == list(g[f"order{ci}_sizes"]), ci
        assert np.array_equal(np.concatenate(idx), g[f"order{ci}_idx"]), ci
        assert between == list(g[f"order{ci}_between"]), ci
        assert len(loader.batch_sampler) == int(g[f"order{ci}_len"]) == len(loader), ci


def test_data_loader_helper_keeps_the_ragged_tail():
    from deepflows_amd.utils.data import data_loader
    g = load_golden("g12_epoch.npz")
    X = np.arange(10, dtype=np.float32).reshape(10, 1)
    np.random.seed(11)
    ld = data_loader(X, np.arange(10), 4, shuffle=True, prefetch_size=0, as_contiguous=True)
    got = [np.asarray(yb) for _, yb in ld]
    assert [len(b) for b in got] == [4, 4, 2]
    assert np.array_equal(np.concatenate(got), g["helper_idx"])
    with pytest.raises(ValueError):
        data_loader(X, np.arange(9), 4)
    assert len(data_loader(X, np.arange(10), 5)) == 2 and len(data_loader(X, np.arange(10), 3)) == 4


def _epoch_setup(g):
    seed, data_seed, n_img, bs, epochs_run, num_epochs = (int(v) for v in g["epoch_meta"])
    rng = np.random.default_rng(data_seed)
    x_all = (rng.standard_normal((n_img, 3, 32, 32)) * 0.9).astype(np.float32)
    y_all = rng.integers(0, 10, n_img)
    return seed, x_all, y_all, bs, epochs_run, num_epochs


def _make_job(dev, seed):
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.optim.scheduler import WarmupCosineLR
    np.random.seed(seed)
    model = workloads.ResNet18(10, device=dev, register_blocks=False)       # the script's list-held blocks (F6)
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
    sched = WarmupCosineLR(opt, warmup_epochs=5, T_max=20, eta_min=1e-5)
    return model, nn.CrossEntropyLoss(), opt, sched


def _check_against_the_reference(g, losses, lrs, model, dev, x_all, tol=1e-4):
    from deepflows_amd.autograd import no_grad
    from deepflows_amd.tensor import Tensor
    assert rel_err(np.asarray(losses, np.float64), g["epoch_losses"]) <= tol
    assert np.allclose(lrs, g["epoch_lrs"], rtol=1e-12, atol=0)
    params = list(model.parameters())
    sums = np.array([float(np.abs(p.data.numpy().astype(np.float64)).sum()) for p in params])
    assert rel_err(sums, g["epoch_param_abs_sum"]) <= tol
    assert rel_err(model.bn1.running_mean.numpy().reshape(-1), g["epoch_running_mean_bn1"].reshape(-1)) <= tol
    model.eval()
    with no_grad():
        logits = model(Tensor(x_all[:2], device=dev)).numpy()
    model.train()
    assert rel_err(logits, g["epoch_eval_logits"]) <= 5e-4          # eval-mode BN through 20 layers of running stats


def test_reference_epoch_loop_on_the_oracle(cpu_oracle):
    """The script's loop, written out with this package's loader / scheduler / model on the oracle device, against
    what the reference's own loop produced: same batches, same augmentation draws, same losses."""
    from deepflows_amd import workloads
    from deepflows_amd.tensor import Graph, Tensor
    from deepflows_amd.utils.data import data_loader
    from deepflows_amd.utils.data_pipeline import draw_decisions
    from oracle import ref_ops
    g = load_golden("g12_epoch.npz")
    seed, x_all, y_all, bs, epochs_run, num_epochs = _epoch_setup(g)
    model, crit, opt, sched = _make_job(cpu_oracle, seed)
    loader = data_loader(x_all, y_all, bs, shuffle=True, prefetch_size=0, as_contiguous=True)
    losses, lrs, idx, erased = [], [], [], []
    for epoch in range(epochs_run):
        model.train()
        for inputs, labels in loader:
            idx.extend(int(np.flatnonzero((x_all == inputs[i]).reshape(len(x_all), -1).all(1))[0]) for i in range(len(inputs)))
            dec = draw_decisions(len(inputs), 32, 32, epoch, num_epochs)
            erased.append(bool(dec[0, 5] > 0))
            xt = Tensor(ref_ops.augment_batch(inputs, dec), device=cpu_oracle)
            yt = Tensor(ref_ops.smooth_onehot(labels, 10, 0.05), device=cpu_oracle)
            loss, _ = workloads.train_step(model, crit, opt, xt, yt)
            losses.append(float(loss.numpy().reshape(-1)[0]))
            Graph.free_graph()
        sched.step()
        lrs.append(opt.lr)
    assert idx == list(g["epoch_idx"])
    assert erased == list(g["epoch_erased"]) and sum(erased) >= 2
    _check_against_the_reference(g, losses, lrs, model, cpu_oracle, x_all)


@pytest.mark.gpu
@pytest.mark.parametrize("tail", ["eager", "capture"])
def test_epoch_trainer_reproduces_the_references_epochs(hip_device, tail):
    from deepflows_amd.utils.data import data_loader
    from deepflows_amd.utils.train_loop import EpochTrainer, train_epoch
    g = load_golden("g12_epoch.npz")
    seed, x_all, y_all, bs, epochs_run, num_epochs = _epoch_setup(g)
    model, crit, opt, sched = _make_job(hip_device, seed)
    loader = data_loader(x_all, y_all, bs, shuffle=True, prefetch_size=0, as_contiguous=True)
    trainer = EpochTrainer(model, crit, opt, hip_device, num_classes=10, eps=0.05, augment=True, capture=True, tail=tail)
    losses, lrs = [], []
    for epoch in range(epochs_run):
        losses.extend(train_epoch(trainer, loader, epoch, num_epochs, scheduler=sched))
        lrs.append(opt.lr)
    assert len(trainer._captured) == (2 if tail == "capture" else 1)
    _check_against_the_reference(g, losses, lrs, model, hip_device, x_all)
    acc = trainer.evaluate(data_loader(x_all, y_all, bs))
    assert 0.0 <= acc <= 100.0
    trainer.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["resnet18", "cnn"])
def test_one_epoch_of_3x256_plus_80_equals_the_eager_loop(hip_device, workload):
    """an "epoch" of 848 images at the benchmark's batch size -- three replays of the captured step
    and the 80-image tail on the eager path -- leaves exactly the parameters of the plain eager loop over the same
    batches, with the batches in the reference's order and the host random numbers (permutation, augmentation
    decisions, Dropout masks) drawn in the reference's order."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Graph, Tensor
    from deepflows_amd.utils.data import data_loader
    from deepflows_amd.utils.data_pipeline import DeviceBatchPipeline
    from deepflows_amd.utils.train_loop import EpochTrainer
    rng = np.random.default_rng(31)
    n = 3 * 256 + 80
    x_all = (rng.standard_normal((n, 3, 32, 32)) * 0.8).astype(np.float32)
    y_all = rng.integers(0, 10, n)
    augment = workload == "resnet18"            # the CNN script trains without augmentation and without smoothing
    eps = 0.05 if augment else 0.0

    def job():
        np.random.seed(3)
        if workload == "resnet18":
            model = workloads.ResNet18(10, device=hip_device, register_blocks=True)
            opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
        else:
            model = workloads.CIFAR10_CNN(10, device=hip_device)
            opt = optim.Adam(model.parameters(), lr=5e-3, weight_decay=5e-4)
        model.train()
        return model, nn.CrossEntropyLoss(), opt

    def finish(model, opt, losses):
        return np.asarray(losses, np.float32), [p.data.numpy().copy() for p in opt.params]

    # (a) the trainer: pipeline look-ahead, captured step for the full batches, eager tail
    model, crit, opt = job()
    trainer = EpochTrainer(model, crit, opt, hip_device, num_classes=10, eps=eps, augment=augment, capture=True, tail="eager")
    np.random.seed(77)
    got_losses = []
    for epoch in range(2):
        got_losses.extend(trainer.run_epoch(data_loader(x_all, y_all, 256, shuffle=True), epoch, 20))
    got = finish(model, opt, got_losses)
    assert len(trainer._captured) == 1
    trainer.destroy()
    Graph.free_graph_all()

    # (b) the plain loop: one batch at a time, fresh tensors, eager steps, the loss read back every batch
    model, crit, opt = job()
    np.random.seed(77)
    want_losses, order = [], []
    for epoch in range(2):
        for xb, yb in data_loader(x_all, y_all, 256, shuffle=True):
            order.append(len(yb))
            pipe = DeviceBatchPipeline(hip_device, xb.shape, num_classes=10, eps=eps, augment=augment)
            xt = Tensor(np.zeros(xb.shape, np.float32), device=hip_device)
            yt = Tensor(np.zeros((len(yb), 10), np.float32), device=hip_device)
            pipe.submit(xb, yb, epoch, 20)
            pipe.deliver(xt, yt)
            loss, _ = workloads.train_step(model, crit, opt, xt, yt)
            want_losses.append(float(loss.numpy().reshape(-1)[0]))
            Graph.free_graph()
    want = finish(model, opt, want_losses)
    assert order == [256, 256, 256, 80] * 2
    assert np.array_equal(got[0], want[0]), (got[0], want[0])
    for a, b in zip(got[1], want[1]):
        assert np.array_equal(a, b)
