"""A training step captured once into a hipGraph and replayed.

The front end issues ~170 kernel launches per ResNet-18 step through Python and ctypes; on an
MI355X those kernels take about as long as the host needs to issue them, so the step is
host-bound.  Everything a step enqueues is shape-static and allocated from the library's
stream-ordered pool, so the whole step (forward, loss, backward, gradient all-reduce, optimizer
update) is captured with hipStreamBeginCapture and replayed with one hipGraphLaunch; blocks the
pool hands out during the capture stay owned by the graph.  The only per-step host state is
Adam's bias correction: the optimizer keeps its scalars in a device array that is refreshed by a
one-thread launch in front of every replay (optim/adam.py: push_hyper / apply).

The reference has nothing comparable (it launches 8 575 device calls per step from Python,
SURVEY F8); this is the MI355X-native replacement for that dispatch loop.
"""
from .tensor import Graph


class CapturedTrainStep:
    """forward + loss + zero_grad + backward (+ optimizer update) captured on `inputs`'s device.
    `inputs` / `targets` are device tensors whose buffers the graph reads on every replay (refill
    them in place to feed new batches); `loss` is the device tensor a replay overwrites.

    With a gradient exchange (`sync`, data parallel) the graph ends after backward: the all-reduce
    (RCCL) and the optimizer kernel are enqueued eagerly behind it on the same stream — three
    launches instead of ~170 — so no collective is ever part of a captured graph."""

    def __init__(self, model, criterion, optimizer, inputs, targets, sync=None):
        mod = inputs.device.mod
        if not (getattr(mod, "fused", False) and hasattr(mod, "capture_begin")):
            raise RuntimeError("CapturedTrainStep needs the hip device")
        if not hasattr(optimizer, "push_hyper"):
            raise RuntimeError("CapturedTrainStep needs an optimizer with device-resident scalars (Adam)")
        self.mod, self.optimizer, self.sync = mod, optimizer, sync
        optimizer.device_hyper = True
        optimizer.push_hyper()                      # allocates the scalar array outside the capture
        mod.capture_begin()
        try:
            outputs = model(inputs)
            loss = criterion(outputs, targets)
            optimizer.zero_grad()
            loss.backward()
            if sync is None:
                optimizer.apply()
            Graph.free_graph()
        except BaseException:
            try:
                mod.capture_end().destroy()
            except Exception:
                pass
            raise
        self.loss, self.outputs = loss, outputs
        self.grads = [p.grad for p in optimizer.params]     # the buffers backward writes on a replay
        self.graph = mod.capture_end(keep=(loss, outputs, inputs, targets, self.grads))
        # the capture only recorded the step: nothing ran, the optimizer's step count is unchanged

    def replay(self):
        opt = self.optimizer
        opt.push_hyper()
        self.graph.replay()
        if self.sync is not None:
            for p, g in zip(opt.params, self.grads):
                p.grad = g
            self.sync()
            opt.apply()
        opt.t += 1
        return self.loss

    def destroy(self):
        self.graph.destroy()
