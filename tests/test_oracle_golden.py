"""CPU tests: the NumPy oracle device + the front-end's primitive path reproduce every golden
vector generated from the reference (oracle/gen_golden.py).  This is what pins the oracle."""
import pytest

import parity_cases as pc


def test_primitives(cpu_oracle):
    pc.check_primitives(cpu_oracle)


@pytest.mark.parametrize("name", pc.CONV_CASES)
def test_conv2d_overwrite(cpu_oracle, name):
    pc.check_conv2d(cpu_oracle, name, "overwrite")


@pytest.mark.parametrize("name", pc.CONV_CASES)
def test_conv2d_accumulate_patch_matches_the_torch_goldens(cpu_oracle, name):
    """oracle/accumulate_mode.py (the `+=` form of the im2col gradient, used as the whole-step oracle of the hip
    device's accumulate mode) against the torch-generated `*_dx_accumulate` vectors."""
    from oracle.accumulate_mode import accumulate_im2col_gradient
    with accumulate_im2col_gradient():
        pc.check_conv2d(cpu_oracle, name, "accumulate")


def test_batchnorm2d(cpu_oracle):
    pc.check_batchnorm(cpu_oracle)


def test_maxpool2d(cpu_oracle):
    pc.check_maxpool(cpu_oracle)


def test_cross_entropy(cpu_oracle):
    pc.check_cross_entropy(cpu_oracle)


def test_linear_relu_reduce(cpu_oracle):
    pc.check_linear_relu_reduce(cpu_oracle)


@pytest.mark.parametrize("cname", list(pc.OPT_CONFIGS))
def test_optimizers(cpu_oracle, cname):
    pc.check_optimizer(cpu_oracle, cname)


def test_linear_regression_trajectory(cpu_oracle):
    pc.check_linear_regression(cpu_oracle)


def test_mlp_three_sgd_steps(cpu_oracle):
    pc.check_mlp(cpu_oracle)


def test_cnn_two_adam_steps(cpu_oracle):
    pc.check_cnn(cpu_oracle)


@pytest.mark.parametrize("mode", ["f6", "all"])
def test_resnet18_one_adam_step(cpu_oracle, mode):
    pc.check_resnet18(cpu_oracle, mode)


@pytest.mark.parametrize("mode", ["f6", "all"])
def test_resnet18_written_as_the_reference_writes_it(cpu_oracle, mode):
    """workloads.ResNet18Plain -- forward() = the text of test/ResNet_CIFAR10_cuda.py:31-47, 84-99 -- against the same golden"""
    pc.check_resnet18(cpu_oracle, mode, plain=True)


def test_ref_ops_match_goldens():
    """oracle/ref_ops.py (the full-size checker) reproduces the conv / BN / pool / CE fixtures."""
    import numpy as np
    from conftest import assert_close, load_golden
    from oracle import ref_ops as R
    g = load_golden("g2_conv2d.npz")
    for name in pc.CONV_CASES:
        N, C, H, W, O, k, s, p = (int(v) for v in g[name + "_geo"])
        b = g[name + "_b"] if (name + "_b") in g else None
        assert_close(R.conv2d_forward(g[name + "_x"], g[name + "_w"], b, s, p), g[name + "_y"], 1e-5, name)
        assert_close(R.conv2d_wgrad(g[name + "_x"], g[name + "_gy"], k, s, p), g[name + "_dw"], 1e-5, name)
        for mode in ("overwrite", "accumulate"):
            assert_close(R.conv2d_dgrad(g[name + "_gy"], g[name + "_w"], (N, C, H, W), s, p, mode),
                         g[name + "_dx_" + mode], 1e-5, name + mode)
    g = load_golden("g3_batchnorm2d.npz")
    y, m, v = R.batchnorm2d_train(g["x1"], g["gamma"], g["beta"])
    assert_close(y, g["y1"], 1e-5, "bn")
    dx, dg, db = R.batchnorm2d_backward(g["gy"], g["x1"], g["gamma"], m, v)
    assert_close(dx, g["dx"], 1e-4, "bn dx"); assert_close(dg, g["dgamma"], 1e-5); assert_close(db, g["dbeta"], 1e-5)
    g = load_golden("g4_maxpool2d.npz")
    y = R.maxpool2d_forward(g["x"], 2)
    assert np.array_equal(y, g["y"]) and np.array_equal(R.maxpool2d_backward(g["x"], y, g["gy"], 2), g["dx"])
    g = load_golden("g5_cross_entropy.npz")
    assert_close(np.array([R.cross_entropy(g["logits"], g["smooth"])]), g["loss_smooth"], 1e-5, "ce")
    g = load_golden("g7_optimizers.npz")
    p_, v_, s_ = g["p0_0"], np.zeros_like(g["p0_0"]), np.zeros_like(g["p0_0"])
    for t in range(3):
        p_, v_, s_ = R.adam_step(p_, g[f"g{t}_0"], v_, s_, t + 1, 1e-3, wd=5e-4)
        assert_close(p_, g[f"adam_wd_step{t + 1}_0"], 1e-6, "adam")


def test_wide_resnet18_one_step(cpu_oracle):
    pc.check_wide_resnet_step(cpu_oracle)


@pytest.mark.parametrize("tag", ["vgg", "mbn"])
def test_vgg16_and_mobilenet_one_step(cpu_oracle, tag):
    """the VGG16 / MobileNetV1 models of test/VGG.py and test/MobileNet.py."""
    pc.check_model_step(cpu_oracle, tag)


def test_avg_pool_activation_modules_and_nll(cpu_oracle):
    """SURVEY §8(f) row 3 on the oracle device: AvgPool2d, LeakyReLU / Tanh / Softmax / LogSoftmax, nll_loss against
    outputs of the reference's own code."""
    pc.check_avg_pool(cpu_oracle)
    pc.check_activation_modules(cpu_oracle)
    pc.check_nll_loss(cpu_oracle)


def test_frozen_parameter_step(cpu_oracle):
    pc.check_frozen_parameter_step(cpu_oracle)
