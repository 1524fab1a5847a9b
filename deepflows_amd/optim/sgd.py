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

    # graph_step.CapturedTrainStep sets device_hyper: the scalars of the update then live in a small device array
    # that push_hyper() rewrites (a one-thread launch in front of the replay) whenever lr / momentum / weight decay /
    # nesterov / grad_scale changed on the host -- a scheduler (optim/scheduler.py mutates optimizer.lr) or a loaded
    # checkpoint takes effect on the next replay instead of being frozen into the captured launch
    device_hyper = False
    _hyper = None
    _hyper_key = None

    def _scalars(self):
        return (float(self.lr), float(self.momentum), float(self.weight_decay), bool(self.nesterov),
                float(self.grad_scale))

    def push_hyper(self):
        mod = self._fused_mod()
        if self._hyper is None:
            self._hyper = mod.Array(16)
        key = self._scalars()
        if key == self._hyper_key:
            return
        if self._hyper_key is not None and (key[1] > 0.0) != (self._hyper_key[1] > 0.0):
            raise RuntimeError("SGD: momentum was switched on or off after the step was captured (the captured launch "
                               "has no momentum buffers / has them): capture the step again")
        if getattr(mod, "capturing", False):
            raise RuntimeError("SGD: the scalars must be written before the capture starts")
        mod.sgd_hyper(self._hyper, *key)
        self._hyper_key = key

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
        if self.device_hyper:
            if self._hyper_key is None or (self._hyper_key != self._scalars() and not getattr(mod, "capturing", False)):
                self.push_hyper()
            mod.sgd_step_dev([self.params[i].data._handle for i in live],
                             [self.params[i].grad.compact()._handle for i in live],
                             [self.v[i]._handle for i in live] if self.momentum > 0.0 else None, self._hyper)
            return
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
