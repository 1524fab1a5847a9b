"""Pooling modules — mirror of DeepFlows/nn/modules/pool.py (MaxPool2d default stride=0,
pool.py:44)."""
from .. import functional as F
from .. import lazy
from .module import Module


class MaxPool2d(Module):
    def __init__(self, kernel_size: int, stride: int = 0, padding: int = 0) -> None:
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    def forward(self, x):
        out = lazy.max_pool2d(self, x)                 # (pool(relu(bn(conv(x)))): the stem as one node, nn/lazy.py)
        if out is not None:
            return out
        return F.max_pool2d(lazy.realize(x), self.kernel_size, self.stride, self.padding)

    def extra_repr(self):
        return f"kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}"


class AvgPool2d(Module):
    def __init__(self, kernel_size: int, stride: int, padding: int = 0) -> None:
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    def forward(self, x):
        return F.avg_pool2d(lazy.realize(x), self.kernel_size, self.stride, self.padding)

    def extra_repr(self):
        return f"kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}"
