"""Optimizer base — mirror of DeepFlows/optim/optimier.py:6-15 (`zero_grad` sets grad=None)."""
from typing import List

from ..tensor import Tensor


class Optimizer:
    def __init__(self, params: List[Tensor]) -> None:
        self.params: List[Tensor] = list(params)
        self.grad_scale = 1.0     # data-parallel: 1/world, applied inside the fused step
        self.t = 0                # steps taken (Adam's bias correction; counted for every optimizer)

    def step(self):
        raise NotImplementedError

    def zero_grad(self):
        for p in self.params:
            p.zero_grad()

    def _fused_mod(self):
        if not self.params:
            return None
        mod = self.params[0].device.mod
        return mod if getattr(mod, "fused", False) else None
