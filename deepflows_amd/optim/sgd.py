"""SGD — mirror of DeepFlows/optim/sgd.py:7-24 (g = grad + p*wd always; momentum
v = v*mu + g; nesterov g + mu*v; p -= lr*update).  On a fused device the whole parameter
list is updated by ONE multi-tensor kernel in place (5-7 launches per parameter in the
reference); the arithmetic and its order are the same."""
from typing import List

from .. import backend_api
from ..tensor import Tensor
from .optimizer import Optimizer


class SGD(Optimizer):
    def __init__(self, params: List[Tensor], lr: float = 1e-2, momentum: float = 0.0,
                 weight_decay: float = 0.0, nesterov: bool = False) -> None:
        super().__init__(params)
        self.lr, self.momentum, self.weight_decay, self.nesterov = lr, momentum, weight_decay, nesterov
        self.v = [backend_api.zeros(p.shape, device=p.device) for p in self.params]

    # graph_step.CapturedTrainStep: SGD has no per-step host state (no bias correction), so its scalars are
    # constants of the captured launch -- a changed learning rate needs a new capture
    device_hyper = False

    def push_hyper(self):
        pass

    def note_replayed_apply(self):
        pass

    def apply(self):
        """The in-place multi-tensor update (one launch for the whole parameter list)."""
        mod = self._fused_mod()
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        if len(live) != len(self.params):
            raise TypeError("SGD.step: a parameter has no gradient")   # as the reference would
        for i in live:
            self.params[i].data = self.params[i].data.compact()
        mod.sgd_step([self.params[i].data._handle for i in live],
                     [self.params[i].grad.compact()._handle for i in live],
                     [self.v[i]._handle for i in live] if self.momentum > 0.0 else None,
                     self.lr, self.momentum, self.weight_decay, self.nesterov, self.grad_scale)

    def step(self):
        if self._fused_mod() is not None:
            self.apply()
            return
        for i, p in enumerate(self.params):
            grad = p.grad if self.grad_scale == 1.0 else p.grad * self.grad_scale
            grad = grad + p.data * self.weight_decay
            if self.momentum > 0.0:
                self.v[i] = self.v[i] * self.momentum + grad
                update = grad + self.momentum * self.v[i] if self.nesterov else self.v[i]
            else:
                update = grad
            p.data -= self.lr * update
