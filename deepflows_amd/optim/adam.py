"""Adam — mirror of DeepFlows/optim/adam.py:7-64: bias-corrected, weight decay added to the
gradient, `t` starts at 1 and advances once per step(), parameters without a gradient are
skipped.  On a fused device: one multi-tensor in-place kernel (17 launches per parameter in
the reference) with the same operation order; scalars are rounded to fp32 where the
reference's NumPy device rounds them."""
import os

from .. import backend_api
from .optimizer import Optimizer


_CHECK_TABLE = os.environ.get("DF_ADAM_CHECK_TABLE", "0") != "0"


class Adam(Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay=0) -> None:
        super().__init__(params)
        self.lr = lr
        self.beta1, self.beta2 = betas
        self.eps = eps
        self.weight_decay = weight_decay
        self.v = [backend_api.zeros_like(p.data) for p in self.params]
        self.s = [backend_api.zeros_like(p.data) for p in self.params]
        self.t = 1

    # ---- fused device ---------------------------------------------------------------------------
    # device_hyper = True (set by graph_step.CapturedTrainStep): the scalars of the update live in a
    # small device array that push_hyper() refreshes and apply() reads, so that apply() can sit in
    # a captured hipGraph while the bias corrections keep advancing from the host.
    device_hyper = False
    _hyper = None

    _HYPER_STEPS = 1024          # steps of scalars kept on the device
    _table_key = None            # (scalars, first step) of the table on the device
    _table_used = 0              # apply() calls since it was written = the index the device is at

    def push_hyper(self):
        """Make sure the device holds the scalars of step self.t at the position its index points to.  The table
        covers _HYPER_STEPS steps and apply() moves the index on by itself, so in steady state this launches
        nothing; it is rewritten when it runs out, when a scalar (lr, weight decay, ...) changed, or when steps
        were taken on another path in between."""
        mod = self._fused_mod()
        if self._hyper is None:
            self._hyper = mod.Array(16 + 16 * self._HYPER_STEPS)
        scalars = (self.lr, self.beta1, self.beta2, self.eps, self.weight_decay, self.grad_scale)
        if (self._table_key is not None and self._table_key[0] == scalars
                and self._table_key[1] + self._table_used == self.t and self._table_used < self._HYPER_STEPS):
            return
        if getattr(mod, "capturing", False):
            raise RuntimeError("Adam: the scalar table must be written before the capture starts")
        if self._table_key is not None and (self._table_used >= self._HYPER_STEPS or _CHECK_TABLE):
            # the table is about to be replaced: what the device consumed must be what this object counted (a replayed
            # step that was not noted, or noted without running, has trained with another step's bias corrections).
            # The read-back drains the device, so it runs when the table is exhausted (once per 1 024 steps) -- a schedule
            # that moves the learning rate every batch rewrites the table in front of every replay and must not wait
            # for the GPU each time -- or on every rewrite with DF_ADAM_CHECK_TABLE=1.
            index, _ = mod.adam_hyper_index(self._hyper)
            if index != self._table_used:
                raise RuntimeError(f"Adam: the device has taken {index} steps from its scalar table, the host counted "
                                   f"{self._table_used}: a captured step was replayed without note_replayed_apply() "
                                   "(or the other way round)")
        mod.adam_hyper_table(self._hyper, self._HYPER_STEPS, self.lr, self.beta1, self.beta2, self.eps,
                             self.weight_decay, self.t, self.grad_scale)
        self._table_key, self._table_used = (scalars, self.t), 0

    def apply(self):
        """The in-place multi-tensor update with the scalars push_hyper() wrote (no host state)."""
        mod = self._fused_mod()
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        if live:
            for i in live:
                self.params[i].data = self.params[i].data.compact()
            mod.adam_step_dev([self.params[i].data._handle for i in live],
                              [self.params[i].grad.compact()._handle for i in live],
                              [self.v[i]._handle for i in live], [self.s[i]._handle for i in live],
                              self._hyper)
            if not getattr(mod, "capturing", False):
                self._table_used += 1            # (a captured apply() runs on replay: note_replayed_apply())

    def note_replayed_apply(self):
        """A graph that contains apply() was replayed once: the device moved its table index on."""
        self._table_used += 1

    def step(self):
        mod = self._fused_mod()
        if mod is not None and self.device_hyper:
            self.push_hyper()
            self.apply()
            self.t += 1
            return
        if mod is not None:
            live = [i for i, p in enumerate(self.params) if p.grad is not None]
            if live:
                for i in live:
                    self.params[i].data = self.params[i].data.compact()
                mod.adam_step([self.params[i].data._handle for i in live],
                              [self.params[i].grad.compact()._handle for i in live],
                              [self.v[i]._handle for i in live], [self.s[i]._handle for i in live],
                              self.lr, self.beta1, self.beta2, self.eps, self.weight_decay, self.t,
                              self.grad_scale)
            self.t += 1
            return
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            g = p.grad.data if hasattr(p.grad, 'data') else p.grad
            if self.grad_scale != 1.0:
                g = g * self.grad_scale
            w = p.data
            if self.weight_decay > 0:
                g = g + w * self.weight_decay
            self.v[i] = self.v[i] * self.beta1 + g * (1 - self.beta1)
            self.s[i] = self.s[i] * self.beta2 + g ** 2 * (1 - self.beta2)
            v_hat = self.v[i] / (1 - self.beta1 ** self.t)
            s_hat = self.s[i] / (1 - self.beta2 ** self.t)
            p.data = w - v_hat / (s_hat ** 0.5 + self.eps) * self.lr
            p.children.clear()
            p.parents.clear()
        self.t += 1
