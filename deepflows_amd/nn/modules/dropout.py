"""Dropout — mirror of DeepFlows/nn/modules/dropout.py:6-36: the mask is drawn on the host
with np.random.binomial (global RNG) and pushed with from_numpy; train: x*mask/(1-p);
eval: x*(1-p) (the reference scales at eval too, dropout.py:31-33 — kept for parity)."""
import numpy as np

from .module import Module


class Dropout(Module):
    def __init__(self, p: float = 0.5):
        super().__init__()
        assert 0 <= p < 1
        self.p = p

    def forward(self, x):
        if self.training:
            mask_np = np.ascontiguousarray(np.random.binomial(1, 1 - self.p, x.shape).astype(np.float32))
            mask = x.device.empty(x.shape, dtype="float32")
            x.device.from_numpy(mask_np, mask._handle)
            return x * mask / (1 - self.p)
        return x * (1 - self.p)

    def extra_repr(self):
        return f"p={self.p}"
