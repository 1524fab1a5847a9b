"""GPU: the BENCHMARKED configurations, end to end, against the oracle (VERDICT r2 "Next" #2).

The end-to-end goldens run at batch 2 / 4 / 8, where none of the kernels the headline number is made of dispatch
(register-operand 3x3 needs N >= 128, the image-per-lane kernels N % 16 == 0, the grouped weight gradient a full
queue).  Here ONE step of exactly `bench.py`'s job (`bench.build_job`: same seeds, same synthetic batch, same
optimizer) runs on the hip device -- eagerly and through `CapturedTrainStep.replay()` -- and on the oracle (NumPy
device + the primitive path = the reference's algorithm, test/ResNet_CIFAR10_cuda.py:20-100,177-202), and loss,
logits, the stem's weight gradient, the per-tensor |grad| sums and the post-step |param| sums are compared under the
measured fp32 budget of tests/error_budget.py (distance from a float64 evaluation of the same algorithm).  The test
also asserts that the fused entry points the bench times were the ones taken (per-family launch counts)."""
import os
import sys

import numpy as np
import pytest

import error_budget as eb
from conftest import ROOT

sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _collect(model, opt, loss, out, params_all):
    res = {"loss": loss.numpy(), "logits": out.numpy(), "conv1_wgrad": params_all[0].grad.numpy(),
           "grad_abs_sum": np.array([float(np.abs(p.grad.numpy().astype(np.float64)).sum()) for p in params_all])}
    return res


def _finish(res, params_all):
    res["param_abs_sum_after"] = np.array([float(np.abs(p.data.numpy().astype(np.float64)).sum()) for p in params_all])
    res["first_param_after"] = params_all[0].data.numpy()
    return {k: np.asarray(v, dtype=np.float64) for k, v in res.items()}


def bench_step(dev, workload, batch, captured=False, counts=None, model_kind="plain"):
    """One step of bench.py's job on `dev`.  Dropout (CNN) draws from the global NumPy generator, which build_job
    seeds: the mask is the same on every device.  model_kind: bench.py's default ("plain": the ResNet-18 written as the
    reference's script writes it) or "fused" (the forward() rewritten around the fused entry points)."""
    import bench
    from deepflows_amd import workloads
    from deepflows_amd.tensor import Graph
    model, opt, crit, xt, yt = bench.build_job(workload, dev, 0, batch, model_kind=model_kind)
    params_all = list(opt.params)
    mod = dev.mod
    if captured:
        from deepflows_amd.graph_step import CapturedTrainStep
        cs = CapturedTrainStep(model, crit, opt, xt, yt)     # records only: the replay draws the Dropout mask
        loss = cs.replay()
        for p, g in zip(opt.params, cs.grads):
            p.grad = g
        res = _collect(model, opt, loss, cs.outputs, params_all)
        out = _finish(res, params_all)
        cs.destroy()
    else:
        if counts is not None:
            mod.prof_reset()
            mod.prof_enable(True)
        outputs = model(xt)
        loss = crit(outputs, yt)
        opt.zero_grad()
        loss.backward()
        res = _collect(model, opt, loss, outputs, params_all)
        opt.step()
        if counts is not None:
            mod.synchronize()
            mod.prof_enable(False)
            counts.update({k: v["calls"] for k, v in mod.prof_read().items()})
        out = _finish(res, params_all)
    Graph.free_graph_all()
    return out


def _oracle(cls, workload, batch, accumulate=False):
    import contextlib
    from oracle.accumulate_mode import accumulate_im2col_gradient
    ctx = accumulate_im2col_gradient() if accumulate else contextlib.nullcontext()
    with ctx:
        return eb.with_device(cls, lambda d: bench_step(d, workload, batch))


CASES = {
    # the headline: ResNet-18 bs 256, overwrite mode, all 62 tensors
    "resnet18_bs256_overwrite": ("resnet18", 256, "overwrite",
                                 {"conv2d_fwd_igemm": 17, "conv2d_wgrad_grouped": 1, "conv2d_dgrad_overwrite_igemm": 16,
                                  # (of the 19 BatchNorm layers behind the stem, the 8 of the 16 x 16 / 8 x 8 stages that do
                                  # not sit in a block with a convolution shortcut and whose gradient comes from such a kernel have no launch of their own)
                                  "bn2d_bwd": 11,
                                  # (5 apply launches -- the bn2 + identity + ReLU of the blocks without a convolution shortcut --
                                  # + 3 for the pairs of a block with one; the eight BatchNorms that stand between two
                                  # convolutions are applied on the second one's loads, and its workgroups add up the first
                                  # one's partial rows themselves: no launch)
                                  # (round 6: the classifier head, the loss and the per-image half of their backward pass are
                                  # ONE launch; dW_fc, db_fc and the mean of the row losses ride on the slab sum: the three
                                  # families of rounds 1-5 -- pool_linear_fwd, softmax_ce_fwd, pool_linear_bwd -- must not run)
                                  "bn2d_fwd": 8, "bn_relu_pool_fwd": 1, "head_ce_fwd_bwd": 1, "wgrad_slab_sum": 1,
                                  "pool_linear_fwd": None, "pool_linear_bwd": None, "softmax_ce_fwd": None,
                                  "head_grads": None, "loss_finish": None, "adam_step": 1}),
    "cnn_bs256_overwrite": ("cnn", 256, "overwrite", {"adam_step": 1}),
    "resnet18_bs128_accumulate": ("resnet18", 128, "accumulate", {"conv2d_wgrad_grouped": 1, "adam_step": 1}),
}


@pytest.mark.parametrize("case", list(CASES))
def test_benchmarked_step_matches_the_oracle(hip_device, case):
    from oracle.numpy_device import NumpyDevice, NumpyDevice64
    workload, batch, dgrad, want_counts = CASES[case]
    hip_device.mod.conv_dgrad_mode = dgrad
    counts = {}
    eager = bench_step(hip_device, workload, batch, counts=counts)
    replayed = bench_step(hip_device, workload, batch, captured=True)
    print("\nlaunch counts per family:", dict(sorted(counts.items())))
    for fam, n in want_counts.items():
        assert counts.get(fam) == n, f"{case}: family {fam} ran {counts.get(fam)} times, the bench's composition has {n}"
    # the replayed graph runs the same kernels in the same order as the eager step
    for k in eager:
        assert np.array_equal(eager[k], replayed[k]), f"{case}: replay differs from the eager step in {k}"
    exact = _oracle(NumpyDevice64, workload, batch, accumulate=(dgrad == "accumulate"))
    fp32 = _oracle(NumpyDevice, workload, batch, accumulate=(dgrad == "accumulate"))
    eb.assert_within_budget(eager, exact, fp32, f"hip {case} (bench.py's job, one step)")
    # the loss bench.py prints for this job is this number
    assert abs(float(eager["loss"].reshape(-1)[0]) - float(fp32["loss"].reshape(-1)[0])) <= 1e-4 * abs(float(fp32["loss"].reshape(-1)[0]))


def test_model_written_as_the_reference_writes_it_launches_the_fused_kernels(hip_device):
    """bench.py's default model (workloads.ResNet18Plain: forward() = the text of test/ResNet_CIFAR10_cuda.py:31-47, 84-99)
    against the class whose forward() was rewritten around the fused entry points: the same launches per family, the same
    bits in every compared quantity."""
    hip_device.mod.conv_dgrad_mode = "overwrite"
    ca, cb = {}, {}
    plain = bench_step(hip_device, "resnet18", 256, counts=ca, model_kind="plain")
    fused = bench_step(hip_device, "resnet18", 256, counts=cb, model_kind="fused")
    print("\nlaunch counts per family (plain):", dict(sorted(ca.items())))
    assert ca == cb, f"different launches: plain {sorted(ca.items())} vs hand-fused {sorted(cb.items())}"
    for k in plain:
        assert np.array_equal(plain[k], fused[k]), f"the plain model differs from the hand-fused one in {k}"


@pytest.mark.parametrize("model_kind", ["plain", "fused"])
def test_eager_steps_return_every_array_to_the_pool_without_the_cycle_collector(hip_device, model_kind):
    """The arrays of a step go back to the library's pool when their Python owners die -- by reference counting: with the
    cycle collector switched off, steady-state eager steps must not reach hipMalloc (a reference cycle among the nodes or
    the gradient tags of a step would keep its arrays until some later collection: the pool then grows by a step's worth of
    memory per step -- tools/soak.py caught exactly that in round 4)."""
    import gc
    import bench
    from deepflows_amd import workloads
    from deepflows_amd.tensor import Graph
    mod = hip_device.mod
    model, opt, crit, xt, yt = bench.build_job("resnet18", hip_device, 0, 64, model_kind=model_kind)
    gc.collect()
    gc.disable()
    try:
        for _ in range(4):
            workloads.train_step(model, crit, opt, xt, yt)
            Graph.free_graph()
        s0 = mod.pool_stats()
        for _ in range(6):
            workloads.train_step(model, crit, opt, xt, yt)
            Graph.free_graph()
        s1 = mod.pool_stats()
    finally:
        gc.enable()
    assert s1["hipmalloc_calls"] == s0["hipmalloc_calls"], (s0, s1)
