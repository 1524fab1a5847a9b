"""Input pipeline: the oracle restatement of augment_batch and the decision
drawing order against outputs of the reference's own function (fixtures from
oracle/gen_augment_golden.py), then the device kernels against the oracle, bit for bit."""
import numpy as np
import pytest

from conftest import load_golden


def _cases():
    g = load_golden("g9_augment.npz")
    for i in range(int(g["cases"])):
        seed, epoch, num_epochs, xseed = (int(v) for v in g[f"meta{i}"])
        x = (np.random.default_rng(xseed).standard_normal((4, 3, 32, 32)) * 0.8).astype(np.float32)
        yield x, g[f"y{i}"], seed, epoch, num_epochs


def test_oracle_and_decision_order_match_the_reference():
    from deepflows_amd.utils.data_pipeline import draw_decisions
    from oracle import ref_ops
    n_erase = 0
    for x, y, seed, epoch, num_epochs in _cases():
        np.random.seed(seed)
        dec = draw_decisions(x.shape[0], x.shape[2], x.shape[3], epoch, num_epochs)
        n_erase += int(dec[0, 5] > 0)
        assert np.array_equal(ref_ops.augment_batch(x, dec), y)
    assert n_erase >= 2                                    # the erase branch is covered
    g = load_golden("g9_augment.npz")
    assert np.array_equal(ref_ops.smooth_onehot(g["labels"], 10, 0.05), g["smoothed"])


@pytest.mark.gpu
def test_device_pipeline_matches_the_reference(hip_device):
    from deepflows_amd import tensor
    from deepflows_amd.utils.data_pipeline import DeviceBatchPipeline
    g = load_golden("g9_augment.npz")
    cases = list(_cases())
    x0 = cases[0][0]
    pipe = DeviceBatchPipeline(hip_device, x0.shape, num_classes=10, eps=0.05)
    xt = tensor.Tensor(np.zeros(x0.shape, np.float32), device=hip_device)
    yt = tensor.Tensor(np.zeros((x0.shape[0], 10), np.float32), device=hip_device)
    labels = g["labels"][:x0.shape[0]]
    # double buffering: two uploads in flight before the first delivery, then strictly alternating
    np.random.seed(cases[0][2])
    pipe.submit(cases[0][0], labels, cases[0][3], cases[0][4])
    for k, (x, y, seed, epoch, num_epochs) in enumerate(cases):
        if k + 1 < len(cases):
            nx = cases[k + 1]
            np.random.seed(nx[2])
            pipe.submit(nx[0], labels, nx[3], nx[4])
        pipe.deliver(xt, yt)
        assert np.array_equal(xt.numpy(), y), k
        assert np.array_equal(yt.numpy(), g["smoothed"][:x0.shape[0]])
