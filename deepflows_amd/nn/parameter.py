"""Parameter = Tensor(requires_grad=True) sharing the source's storage
(mirror of DeepFlows/nn/parameter.py:5-34)."""
from .. import backend_api
from ..tensor import Tensor


class Parameter(Tensor):
    def __init__(self, data: Tensor):
        super().__init__(array=data.data, dtype=data.dtype, device=data.device, requires_grad=True)

    def to(self, device):
        dev = backend_api.Device(device) if isinstance(device, str) else device
        if self.device == dev:
            return self
        return self.__class__(Tensor(self.data.numpy(), dtype=self.dtype, device=dev))

    def __repr__(self):
        return f"Parameter : \n{self.data},\ndevice={self.device}"
