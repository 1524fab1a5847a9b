"""Initialisers — mirror of DeepFlows/nn/init.py.  Values are drawn on the host from the
GLOBAL NumPy RNG in call order and pushed with `device.from_numpy(float64_array, handle)`
(init.py:13-24), so a seeded model construction reproduces the reference's weights bit for
bit (fan_in = shape[1]; kaiming_uniform_ keeps the default nonlinearity 'relu' whatever `a`
is, init.py:139-152)."""
import math
import warnings

import numpy as np
from numpy.random import normal, uniform

from ..autograd import no_grad
from ..tensor import Tensor


def _push(tensor, values):
    with no_grad():
        tensor.data.device.from_numpy(np.ascontiguousarray(values), tensor.data._handle)
    return tensor


def _no_grad_uniform_(tensor, a, b):
    return _push(tensor, uniform(a, b, tensor.shape))


def _no_grad_normal_(tensor, mean, std):
    return _push(tensor, normal(mean, std, size=tensor.shape))


def _no_grad_fill_(tensor, val):
    with no_grad():
        tensor.data.fill(val)
    return tensor


def calculate_gain(nonlinearity, param=None):
    if nonlinearity in ('linear', 'conv1d', 'conv2d', 'conv3d', 'conv_transpose1d',
                        'conv_transpose2d', 'conv_transpose3d', 'sigmoid'):
        return 1
    if nonlinearity == 'tanh':
        return 5.0 / 3
    if nonlinearity == 'relu':
        return math.sqrt(2.0)
    if nonlinearity == 'leaky_relu':
        if param is None:
            slope = 0.01
        elif (not isinstance(param, bool) and isinstance(param, int)) or isinstance(param, float):
            slope = param
        else:
            raise ValueError(f"negative_slope {param} not a valid number")
        return math.sqrt(2.0 / (1 + slope ** 2))
    if nonlinearity == 'selu':
        return 3.0 / 4
    raise ValueError(f"Unsupported nonlinearity {nonlinearity}")


def _calculate_fan_in_and_fan_out(tensor: Tensor):
    if tensor.ndim < 2:
        raise ValueError("Fan in and fan out can not be computed for tensor with fewer than 2 dimensions")
    receptive = math.prod(tensor.shape[2:]) if tensor.ndim > 2 else 1
    return tensor.shape[1] * receptive, tensor.shape[0] * receptive


def _calculate_correct_fan(tensor, mode):
    mode = mode.lower()
    if mode not in ('fan_in', 'fan_out'):
        raise ValueError(f"Mode {mode} not supported, please use one of ['fan_in', 'fan_out']")
    fan_in, fan_out = _calculate_fan_in_and_fan_out(tensor)
    return fan_in if mode == 'fan_in' else fan_out


def normal_(tensor, mean=0., std=1.):
    return _no_grad_normal_(tensor, mean, std)


def uniform_(tensor, low=0., high=1.0):
    return _no_grad_uniform_(tensor, low, high)


def fill_(tensor, val):
    return _no_grad_fill_(tensor, val)


def zeros_(tensor):
    return _no_grad_fill_(tensor, 0.)


def ones_(tensor):
    return _no_grad_fill_(tensor, 1.)


def xavier_uniform_(tensor, gain=1.0):
    fan_in, fan_out = _calculate_fan_in_and_fan_out(tensor)
    bound = gain * math.sqrt(6. / (fan_in + fan_out))
    return _no_grad_uniform_(tensor, -bound, bound)


def xavier_normal_(tensor, gain=1.0):
    fan_in, fan_out = _calculate_fan_in_and_fan_out(tensor)
    return _no_grad_normal_(tensor, 0., gain * math.sqrt(2.0 / (fan_in + fan_out)))


def kaiming_uniform_(tensor, a=0, mode="fan_in", nonlinearity="relu"):
    if 0 in tensor.shape:
        warnings.warn("Initializing zero-element tensors is a no-op")
        return tensor
    std = calculate_gain(nonlinearity, a) / math.sqrt(_calculate_correct_fan(tensor, mode))
    bound = math.sqrt(3.0) * std
    return _no_grad_uniform_(tensor, -bound, bound)


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    if 0 in tensor.shape:
        warnings.warn("Initializing zero-element tensors is a no-op")
        return tensor
    std = calculate_gain(nonlinearity, a) / math.sqrt(_calculate_correct_fan(tensor, mode))
    return _no_grad_normal_(tensor, 0, std)
