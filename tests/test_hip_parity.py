"""GPU parity tests (pytest -m gpu): the `hip` device — primitives and fused kernels, all
called through the C ABI of libdfhip.so — against the golden vectors generated from the
reference, with the same assertions the oracle passes on CPU."""
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu


def test_device_is_the_native_extension(hip_device):
    import deepflows_amd._lib as L
    assert hip_device.name == "hip" and hip_device.mod.fused
    assert L._lib is not None, "libdfhip.so must be the thing that runs"
    assert "gfx950" in hip_device.mod.device_name, hip_device.mod.device_name


def test_primitives(hip_device):
    pc.check_primitives(hip_device)


@pytest.mark.parametrize("name", pc.CONV_CASES)
def test_conv2d_overwrite(hip_device, name):
    pc.check_conv2d(hip_device, name, "overwrite")


@pytest.mark.parametrize("name", pc.CONV_CASES)
def test_conv2d_accumulate(hip_device, name):
    pc.check_conv2d(hip_device, name, "accumulate")


def test_batchnorm2d(hip_device):
    pc.check_batchnorm(hip_device)


def test_maxpool2d(hip_device):
    pc.check_maxpool(hip_device)


def test_cross_entropy(hip_device):
    pc.check_cross_entropy(hip_device)


def test_linear_relu_reduce(hip_device):
    pc.check_linear_relu_reduce(hip_device)


@pytest.mark.parametrize("cname", list(pc.OPT_CONFIGS))
def test_optimizers(hip_device, cname):
    pc.check_optimizer(hip_device, cname)


def test_linear_regression_trajectory(hip_device):
    pc.check_linear_regression(hip_device)


def test_mlp_three_sgd_steps(hip_device):
    pc.check_mlp(hip_device)


def test_cnn_two_adam_steps(hip_device):
    pc.check_cnn(hip_device)


@pytest.mark.parametrize("mode", ["f6", "all"])
def test_resnet18_one_adam_step(hip_device, mode):
    pc.check_resnet18(hip_device, mode)
