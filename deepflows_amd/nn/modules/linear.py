"""Linear — mirror of DeepFlows/nn/modules/linear.py:10-67: weight stored (in, out), bias
(1, out); y = x @ W + b."""
import math

from ... import backend_api
from ...tensor import Tensor, empty
from .. import functional as F
from .. import lazy
from .. import init
from ..parameter import Parameter
from .module import Module


class Linear(Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, device="hip",
                 dtype="float32") -> None:
        super().__init__()
        kw = {'device': backend_api.Device(device), 'dtype': dtype}
        self.in_features, self.out_features = in_features, out_features
        self.weight = Parameter(empty((in_features, out_features), **kw))
        if bias:
            self.bias = Parameter(empty((1, out_features), **kw))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)   # = out_features
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input: Tensor) -> Tensor:
        out = lazy.linear(self, input)                 # (fc(mean(mean(x, 2), 2)): the classifier head as one node, nn/lazy.py)
        if out is not None:
            return out
        return F.linear(lazy.realize(input), self.weight, self.bias)

    def extra_repr(self) -> str:
        return f'in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}'
