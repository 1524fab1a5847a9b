"""Conv2d — mirror of DeepFlows/nn/modules/conv.py:69-124: weight (O,C,k,k), bias (1,O,1,1),
kaiming_uniform_(a=sqrt(5)) then uniform bias in +-1/sqrt(fan_in), drawn in that order."""
import math

from ... import backend_api, tensor
from .. import functional as F
from .. import lazy
from .. import init
from ..parameter import Parameter
from .module import Module


class Conv2d(Module):
    def __init__(self, in_channels, out_channels, kernel_size, padding=0, stride=1, bias=True,
                 device="hip", dtype=None) -> None:
        super().__init__()
        kw = {"device": backend_api.Device(device), "dtype": dtype}
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.padding, self.stride = kernel_size, padding, stride
        self.weight = Parameter(tensor.empty((out_channels, in_channels, kernel_size, kernel_size), **kw))
        self.bias = Parameter(tensor.empty((1, out_channels, 1, 1), **kw)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            if fan_in != 0:
                bound = 1 / math.sqrt(fan_in)
                init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        # (on a fused device the call may be recorded instead of launched: nn/lazy.py)
        return lazy.conv2d(self, x)

    def extra_repr(self):
        return (f"in_channels={self.in_channels}, out_channels={self.out_channels}, "
                f"kernel_size={self.kernel_size}, padding={self.padding}, stride={self.stride}, "
                f"bias={self.bias is not None}")
