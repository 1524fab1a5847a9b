"""CPU tests: the NumPy oracle device + the front-end's primitive path reproduce every golden
vector generated from the reference (oracle/gen_golden.py).  This is what pins the oracle."""
import pytest

import parity_cases as pc


def test_primitives(cpu_oracle):
    pc.check_primitives(cpu_oracle)


@pytest.mark.parametrize("name", pc.CONV_CASES)
def test_conv2d_overwrite(cpu_oracle, name):
    pc.check_conv2d(cpu_oracle, name, "overwrite")


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
