"""Where an end-to-end comparison needs more than the 1e-4 contract, the allowance is MEASURED: the reference's
algorithm is evaluated in float64 (oracle NumpyDevice64) and every quantity is judged by its distance from that,
against what the float32 oracle (= the reference's own arithmetic) loses on the same path (tests/error_budget.py)."""
import numpy as np
import pytest

import error_budget as eb
from conftest import load_golden, rel_err


def _r18_golden(mode="all"):
    g = load_golden("g8_end_to_end.npz")
    return {"loss": g[f"r18_{mode}_loss"], "logits": g[f"r18_{mode}_logits"], "fc_wgrad": g[f"r18_{mode}_fc_wgrad"],
            "conv1_wgrad": g[f"r18_{mode}_conv1_wgrad"], "grad_abs_sum": g[f"r18_{mode}_grad_abs_sum"],
            "param_abs_sum_after": g[f"r18_{mode}_param_abs_sum_after"]}


def test_fp32_error_of_the_resnet18_step_is_measured():
    """CPU: fp64 vs fp32 oracle vs the REFERENCE's own fp32 outputs (the golden).  The reference's own result sits
    as far from the exact one as the oracle's: the golden comparisons at the end of the conv chain cannot be held to
    1e-4, and 2 x this measured distance is what the GPU test allows instead."""
    from oracle.numpy_device import NumpyDevice, NumpyDevice64
    exact = eb.with_device(NumpyDevice64, eb.resnet18_step)
    fp32 = eb.with_device(NumpyDevice, eb.resnet18_step)
    gold = _r18_golden()
    print("\nResNet-18 batch-2 step, relative distance from the float64 evaluation:")
    print(f"  {'quantity':24s} {'fp32 oracle':>12s} {'reference (golden)':>20s}")
    for k in gold:
        e32, eg = rel_err(fp32[k], exact[k]), rel_err(np.asarray(gold[k], np.float64).reshape(exact[k].shape), exact[k])
        print(f"  {k:24s} {e32:12.3e} {eg:20.3e}")
        assert e32 < 5e-3 and eg < 5e-3, k            # fp32 is still fp32
        # the oracle restates the reference: both lose about the same on every quantity
        assert eg <= max(1e-4, 4.0 * e32) and e32 <= max(1e-4, 4.0 * eg), k
    # forward quantities are inside the plain contract
    assert rel_err(fp32["loss"], exact["loss"]) < 1e-4 and rel_err(fp32["logits"], exact["logits"]) < 1e-4


@pytest.mark.gpu
def test_hip_resnet18_step_within_the_measured_budget(hip_device):
    from oracle.numpy_device import NumpyDevice, NumpyDevice64
    exact = eb.with_device(NumpyDevice64, eb.resnet18_step)
    fp32 = eb.with_device(NumpyDevice, eb.resnet18_step)
    got = eb.resnet18_step(hip_device)
    eb.assert_within_budget(got, exact, fp32, "hip ResNet-18 step (batch 2, overwrite dgrad, all 62 tensors)")


@pytest.mark.gpu
def test_hip_cnn_steps_within_the_measured_budget(hip_device):
    from oracle.numpy_device import NumpyDevice, NumpyDevice64
    exact = eb.with_device(NumpyDevice64, eb.cnn_steps)
    fp32 = eb.with_device(NumpyDevice, eb.cnn_steps)
    got = eb.cnn_steps(hip_device)
    eb.assert_within_budget(got, exact, fp32, "hip CNN, two Adam steps (batch 4)")


def test_fp32_error_of_the_cnn_steps_is_measured():
    from oracle.numpy_device import NumpyDevice, NumpyDevice64
    exact = eb.with_device(NumpyDevice64, eb.cnn_steps)
    fp32 = eb.with_device(NumpyDevice, eb.cnn_steps)
    g = load_golden("g8_end_to_end.npz")
    gold = {"logits0": g["cnn_logits0"], "gnorms0": g["cnn_gnorms0"], "conv1_wgrad0": g["cnn_conv1_wgrad0"],
            "losses": g["cnn_losses"]}
    print("\nCNN, two Adam steps, relative distance from the float64 evaluation:")
    for k in gold:
        e32, eg = rel_err(fp32[k], exact[k]), rel_err(np.asarray(gold[k], np.float64).reshape(exact[k].shape), exact[k])
        print(f"  {k:16s} fp32 oracle {e32:10.3e}   reference (golden) {eg:10.3e}")
        assert e32 < 5e-3 and eg < 5e-3, k
