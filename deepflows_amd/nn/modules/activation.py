"""Activation modules — mirror of DeepFlows/nn/modules/activation.py (the functional ones
that work in the reference: ReLU, Tanh, LeakyReLU, Softmax, LogSoftmax)."""
from .. import functional as F
from .. import lazy
from .module import Module

__all__ = ['ReLU', 'Tanh', 'LeakyReLU', 'Softmax', 'LogSoftmax']


class ReLU(Module):
    def forward(self, input):
        return F.relu(input)                           # (on a recorded call F.relu extends / completes the record: nn/lazy.py)


class Tanh(Module):
    def forward(self, input):
        return F.tanh(input)


class LeakyReLU(Module):
    def __init__(self, negative_slope: float = 1e-2) -> None:
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, self.negative_slope)


class Softmax(Module):
    def __init__(self, dim=None) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, input):
        return F.softmax(input, self.dim)


class LogSoftmax(Module):
    def __init__(self, dim=None) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, input):
        return F.log_softmax(input, self.dim)
