"""Measured error budgets for the end-to-end parity checks.

The fp32 contract is 1e-4 relative (max|a-b| / max|b|).  At the end of a 20-convolution chain two CORRECT fp32
evaluations of the same algorithm that only differ in summation order sit further apart than that, so a flat
tolerance there is either too loose or fails on the reference itself.  These helpers run the reference's algorithm
in float64 (oracle.numpy_device.NumpyDevice64: same front end, same primitive path, same overwrite semantics) and
express every comparison as distances from that exact result:

    err32  = |fp32 oracle (the reference's own arithmetic) - fp64|     what float32 itself costs
    err    = |device under test - fp64|

and the assertion is  err <= max(1e-4, 2 * err32)  per quantity -- never a flat number above 1e-4.  Round 4: where the fp32
oracle is itself stable (err32 <= 1e-5) the distance the north star names is asserted as well, flat:
|device under test - fp32 oracle| <= 1e-4 relative; it is printed for every quantity."""
import numpy as np

from conftest import load_golden, rel_err


def resnet18_step(dev, mode="all", dgrad_mode=None):
    """One Adam step of ResNet-18 (CIFAR variant) at batch 2 on the g8 golden inputs; returns the quantities the
    parity tests look at, as float64 arrays."""
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Graph, Tensor
    g = load_golden("g8_end_to_end.npz")
    np.random.seed(0)
    model = workloads.ResNet18(10, device=dev, register_blocks=(mode == "all"))
    model.train()
    params_all = workloads.all_block_parameters(model)
    opt = optim.Adam(list(model.parameters()), lr=1e-3, weight_decay=5e-4)
    out = model(Tensor(g["r18_x"], device=dev))
    loss = nn.CrossEntropyLoss()(out, Tensor(g["r18_y"], device=dev))
    opt.zero_grad()
    loss.backward()
    res = {"loss": loss.numpy(), "logits": out.numpy(), "fc_wgrad": model.fc.weight.grad.numpy(),
           "conv1_wgrad": model.conv1.weight.grad.numpy(),
           "grad_abs_sum": np.array([float(np.abs(p.grad.numpy().astype(np.float64)).sum()) for p in params_all])}
    opt.step()
    res["param_abs_sum_after"] = np.array([float(np.abs(p.data.numpy().astype(np.float64)).sum()) for p in params_all])
    res["conv1_w_after"] = model.conv1.weight.data.numpy()
    Graph.free_graph_all()
    return {k: np.asarray(v, dtype=np.float64) for k, v in res.items()}


def wide_resnet_step(dev):
    """One SGD step of test/ResNet.py's ResNet-18 (64/128/256/512, ReLU after the add) at batch 2 on the g10 inputs."""
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Graph, Tensor
    g = load_golden("g10_models.npz")
    np.random.seed(0)
    model = workloads.ResNet18Wide(10, device=dev, register_blocks=False)
    model.train()
    every = workloads.all_block_parameters(model)
    opt = optim.SGD(list(model.parameters()), lr=0.01, momentum=0.9, weight_decay=0.0005)
    out = model(Tensor(g["x"], device=dev))
    loss = nn.CrossEntropyLoss()(out, Tensor(g["y"], device=dev))
    opt.zero_grad()
    loss.backward()
    res = {"loss": loss.numpy(), "logits": out.numpy(), "fc_wgrad": model.fc.weight.grad.numpy(),
           "conv1_wgrad": model.conv1.weight.grad.numpy(),
           "grad_abs_sum": np.array([float(np.abs(p.grad.numpy().astype(np.float64)).sum()) for p in every])}
    opt.step()
    res["param_abs_sum_after"] = np.array([float(np.abs(p.data.numpy().astype(np.float64)).sum()) for p in every])
    Graph.free_graph_all()
    return {k: np.asarray(v, dtype=np.float64) for k, v in res.items()}


def cnn_steps(dev):
    """Two Adam steps of the CNN workload on the g8 golden inputs (Dropout masks from the seeded global generator)."""
    from deepflows_amd import nn, optim, workloads
    from deepflows_amd.tensor import Graph, Tensor
    g = load_golden("g8_end_to_end.npz")
    np.random.seed(0)
    model = workloads.CIFAR10_CNN(10, device=dev)
    model.train()
    opt = optim.Adam(model.parameters(), lr=0.005, weight_decay=5e-4)
    crit = nn.CrossEntropyLoss()
    res, losses = {}, []
    for step in range(2):
        out = model(Tensor(g["cnn_x"], device=dev))
        loss = crit(out, Tensor(g["cnn_y"], device=dev))
        opt.zero_grad()
        loss.backward()
        if step == 0:
            res["logits0"] = out.numpy()
            res["gnorms0"] = np.array([np.linalg.norm(p.grad.numpy().astype(np.float64)) for p in model.parameters()])
            res["conv1_wgrad0"] = model.conv1.weight.grad.numpy()
        opt.step()
        losses.append(float(loss.numpy().reshape(-1)[0]))
        Graph.free_graph()
    res["losses"] = np.array(losses)
    Graph.free_graph_all()
    return {k: np.asarray(v, dtype=np.float64) for k, v in res.items()}


def with_device(cls, fn):
    """Run fn(device) on a freshly registered oracle device of class `cls` (NumpyDevice / NumpyDevice64)."""
    from deepflows_amd import backend_api
    from deepflows_amd.tensor import Graph
    name = "cpu_budget"
    dev = backend_api.register_device(name, cls())
    try:
        return fn(dev)
    finally:
        Graph.free_graph_all()
        backend_api.unregister_device(name)


def budget_table(got, exact, fp32):
    """[(name, err(got, exact), err(fp32, exact), allowed, dist(got, fp32 oracle), its limit or None)]: allowed =
    max(1e-4, 2 * err32) against the float64 evaluation; and, where the fp32 oracle itself sits within STABLE of the exact
    result (no tie / ReLU-mask flips between two fp32 evaluations: loss, logits, |grad| sums, |param| sums), the distance
    the north star names -- device under test against the reference's own fp32 arithmetic -- with the flat 1e-4."""
    rows = []
    for k in exact:
        e32 = rel_err(fp32[k], exact[k])
        direct = rel_err(got[k], fp32[k])
        rows.append((k, rel_err(got[k], exact[k]), e32, max(1e-4, 2.0 * e32), direct, 1e-4 if e32 <= STABLE else None))
    return rows


STABLE = 1e-5


def assert_within_budget(got, exact, fp32, what):
    rows = budget_table(got, exact, fp32)
    print(f"\n{what}: distance from the float64 evaluation of the reference's algorithm, and from its fp32 evaluation")
    print(f"  {'quantity':24s} {'under test':>12s} {'fp32 oracle':>12s} {'allowed':>12s} {'vs fp32 oracle':>15s}  asserted")
    for k, e, e32, lim, direct, dlim in rows:
        note = "<= 1e-4" if dlim is not None else "fp64 budget only: the fp32 oracle is itself %.1e from exact (ties / ReLU masks)" % e32
        print(f"  {k:24s} {e:12.3e} {e32:12.3e} {lim:12.3e} {direct:15.3e}  {note}")
    bad = [(k, e, lim) for k, e, _, lim, _, _ in rows if e > lim]
    assert not bad, f"{what}: outside max(1e-4, 2 x the fp32 oracle's own error): {bad}"
    bad = [(k, direct) for k, _, _, _, direct, dlim in rows if dlim is not None and direct > dlim]
    assert not bad, f"{what}: more than 1e-4 from the reference's fp32 evaluation: {bad}"
