"""Dropout — mirror of DeepFlows/nn/modules/dropout.py:6-36: the mask is drawn on the host
with np.random.binomial (global RNG) and pushed with from_numpy; train: x*mask/(1-p);
eval: x*(1-p) (the reference scales at eval too, dropout.py:31-33 — kept for parity)."""
import numpy as np

from .module import Module


def _continue_numpy_stream(keep, shape):
    """The mask from libdfhip's host helper (NumPy's MT19937 stream continued in vectorised C, the global generator
    advanced exactly as NumPy would have advanced it); None when the library or the generator kind does not allow it."""
    import ctypes
    try:
        from ... import _lib
        if not _lib.library_present():
            return None
        lib = _lib.load()
    except Exception:
        return None
    st = np.random.get_state()
    if st[0] != "MT19937":
        return None
    n = 1
    for v in shape:
        n *= int(v)
    key = np.array(st[1], dtype=np.uint32, copy=True)
    pos = ctypes.c_int(int(st[2]))
    out = np.empty(n, np.float32)
    rc = lib.dfhip_host_bernoulli_mask(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), float(keep), n,
                                       out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        return None
    np.random.set_state(("MT19937", key, pos.value, st[3], st[4]))
    return out.reshape(tuple(int(v) for v in shape))


def bernoulli_mask(keep, shape):
    """np.random.binomial(1, keep, shape) as float32 — the same values and the same consumption of the
    global generator (one double per element: NumPy's inversion sampler for n = 1 reduces to a
    threshold on that double, mirrored for keep > 0.5).  Large masks (the CNN workload draws 0.5 M elements per
    step and its replayed step was bound by exactly that) continue the generator's stream in libdfhip's host helper,
    about three times faster than NumPy's scalar generator; tests/test_frontend_host_logic.py pins both
    equivalences."""
    if int(np.prod([int(v) for v in shape])) >= 8192:
        m = _continue_numpy_stream(keep, shape)
        if m is not None:
            return m
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
                up = getattr(x.device.mod, "upload_bernoulli_mask", None)
                if up is None or mask._handle.size < 8192 or not up(mask._handle, 1 - self.p):
                    mask_np = np.ascontiguousarray(bernoulli_mask(1 - self.p, x.shape))
                    x.device.from_numpy(mask_np, mask._handle)
            from .. import functional as F, lazy
            out = lazy.dropout(self, x, mask, 1 - self.p)      # (fc(drop(x)) + CrossEntropyLoss: one launch, nn/lazy.py)
            if out is not None:
                return out
            return F.dropout_train(x, mask, 1 - self.p)
        return x * (1 - self.p)

    def extra_repr(self):
        return f"p={self.p}"
