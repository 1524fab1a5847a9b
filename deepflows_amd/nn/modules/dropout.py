"""Dropout — mirror of DeepFlows/nn/modules/dropout.py:6-36: the mask is drawn on the host
with np.random.binomial (global RNG) and pushed with from_numpy; train: x*mask/(1-p);
eval: x*(1-p) (the reference scales at eval too, dropout.py:31-33 — kept for parity)."""
import numpy as np

from .module import Module


def bernoulli_mask(keep, shape):
    """np.random.binomial(1, keep, shape) as float32 — the same values and the same consumption of the
    global generator (one double per element: NumPy's inversion sampler for n = 1 reduces to a
    threshold on that double, mirrored for keep > 0.5), ten times faster for the 0.5 M-element masks
    of the CNN workload (tests/test_frontend_host_logic.py pins the equivalence)."""
    u = np.random.random_sample(tuple(int(s) for s in shape))
    m = (u > 1.0 - keep) if keep <= 0.5 else ~(u > keep)
    return m.astype(np.float32)


class Dropout(Module):
    def __init__(self, p: float = 0.5):
        super().__init__()
        assert 0 <= p < 1
        self.p = p

    def forward(self, x):
        if self.training:
            mask = x.device.empty(x.shape, dtype="float32")
            feeds = getattr(x.device.mod, "capture_feeds", None)
            if getattr(x.device.mod, "capturing", False) and feeds is not None:
                # inside a step capture (graph_step.CapturedTrainStep) nothing runs and no host copy may be
                # recorded: the mask array belongs to the graph and is a graph INPUT -- every replay draws the
                # masks in this registration order (= the forward order = the reference's RNG order,
                # dropout.py:16-18) and uploads them in front of the graph launch
                feeds.append((mask, 1 - self.p))
            else:
                mask_np = np.ascontiguousarray(bernoulli_mask(1 - self.p, x.shape))
                x.device.from_numpy(mask_np, mask._handle)
            return x * mask / (1 - self.p)
        return x * (1 - self.p)

    def extra_repr(self):
        return f"p={self.p}"
