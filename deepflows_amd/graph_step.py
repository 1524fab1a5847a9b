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
import os
import sys

import numpy as np

from .tensor import Graph


class CapturedTrainStep:
    """forward + loss + zero_grad + backward (+ optimizer update) captured on `inputs`'s device.
    `inputs` / `targets` are device tensors whose buffers the graph reads on every replay (refill
    them in place to feed new batches); `loss` is the device tensor a replay overwrites.

    With a gradient exchange (`sync`, data parallel) the all-reduce (RCCL) and the optimizer kernel are part of the
    captured graph where the collective can be captured (`exchange_captured`), else the graph ends after backward and they
    are enqueued eagerly behind every replay on the same stream."""

    def __init__(self, model, criterion, optimizer, inputs, targets, sync=None, capture_exchange=None, steps_per_graph=1):
        """steps_per_graph = S > 1 (round 6): S consecutive steps in ONE graph -- a graph launch costs 5-8 us of device time
        whatever it holds (tools/probes/queue_probe.hip: 12.9 us for a graph of three empty kernels, 107 us for 58), a fifth of
        the perceptron's 19-us step.  `replay()` still advances ONE step per call: the graph is launched on every S-th call
        and holds that call's step and the S - 1 that follow (the loss it returns is the one of the graph's last step;
        Dropout masks of all S steps are drawn in step order in front of the launch; a tail of fewer than S calls has run
        the whole graph).  `inputs` / `targets` may be lists of S tensors (one batch per step of the graph)."""
        self.S = int(steps_per_graph)
        self._phase = 0
        if self.S not in (1, 2, 4, 8) or (self.S > 1 and sync is not None):
            # (a power of two: Adam's device table of 1 024 steps' scalars is rewritten in front of a launch, never inside a graph)
            raise ValueError("steps_per_graph: 1, 2, 4 or 8, and 1 with a gradient exchange")
        self._inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs] * self.S
        self._targets = list(targets) if isinstance(targets, (list, tuple)) else [targets] * self.S
        if len(self._inputs) != self.S or len(self._targets) != self.S:
            raise ValueError("steps_per_graph: one input / target tensor, or a list of steps_per_graph of them")
        inputs, targets = self._inputs[0], self._targets[0]
        mod = inputs.device.mod
        if not (getattr(mod, "fused", False) and hasattr(mod, "capture_begin")):
            raise RuntimeError("CapturedTrainStep needs the hip device")
        if not hasattr(optimizer, "push_hyper"):
            raise RuntimeError("CapturedTrainStep needs an optimizer with device-resident scalars (Adam)")
        self.mod, self.optimizer, self.sync = mod, optimizer, sync
        optimizer.device_hyper = True
        optimizer.push_hyper()                      # allocates the scalar array outside the capture
        # Round 4: the exchange step is captured too where the collective library lets a stream capture record it (RCCL
        # does for ncclAllReduce): a data-parallel step is then ONE graph launch -- backward, all-reduce of the bucket the
        # backward kernels wrote, optimizer -- with nothing issued eagerly behind it.  If the capture refuses (an older
        # librccl, a collective that needs host interaction), the step is captured again without it and the exchange is
        # enqueued behind every replay as in rounds 1-3; `exchange_captured` / `exchange_fallback_reason` say which.
        if capture_exchange is None:
            capture_exchange = os.environ.get("DF_CAPTURE_EXCHANGE", "1") != "0"
        self.exchange_captured, self.exchange_fallback_reason = False, None
        if sync is not None and capture_exchange:
            try:
                self._capture(model, criterion, optimizer, inputs, targets, with_exchange=True)
                self.exchange_captured = True
                return
            except Exception as e:                  # (the failed capture has been closed and destroyed)
                self.exchange_fallback_reason = f"{type(e).__name__}: {e}"
                print(f"[graph_step] the gradient exchange could not be captured ({self.exchange_fallback_reason}); "
                      "it is enqueued behind every replay instead", file=sys.stderr)
        self._capture(model, criterion, optimizer, inputs, targets, with_exchange=False)

    def _capture(self, model, criterion, optimizer, inputs, targets, with_exchange):
        mod, sync = self.mod, self.sync
        mod.capture_feeds = []                      # host-drawn inputs of the step (Dropout masks), in forward order
        mod.capture_begin()
        try:
            for s_i in range(self.S):
                outputs = model(self._inputs[s_i])
                loss = criterion(outputs, self._targets[s_i])
                optimizer.zero_grad()
                loss.backward()
                if sync is not None and with_exchange:
                    sync()
                if sync is None or with_exchange:
                    optimizer.apply()
                Graph.free_graph()
        except BaseException:
            mod.capture_feeds = None
            try:
                mod.capture_end().destroy()
            except Exception:
                pass
            Graph.free_graph()
            raise
        self.feeds, mod.capture_feeds = mod.capture_feeds, None
        self.loss, self.outputs = loss, outputs
        self.grads = [p.grad for p in optimizer.params]     # the buffers backward writes on a replay
        self.graph = mod.capture_end(keep=(loss, outputs, self._inputs, self._targets, self.grads, self.feeds))
        # the capture only recorded the step: nothing ran, the optimizer's step count is unchanged

    def replay(self):
        opt = self.optimizer
        if self.S > 1:
            # one step per call: the graph of S steps goes out on the first call of every S
            self._phase = (self._phase + 1) % self.S
            if self._phase != 1 % self.S:
                opt.note_replayed_apply()
                opt.t += 1
                return self.loss
        opt.push_hyper()
        if self.feeds and os.environ.get("DF_DEBUG_SKIP_FEEDS") != "1":     # (diagnostic switch: timing without the uploads)
            # Dropout masks of this step: drawn on the host exactly as the eager forward pass would draw them
            # (nn/modules/dropout.py); they cross PCIe beside the step that is running and become the graph's input arrays
            # with one launch in front of the graph (dfhip_upload_bernoulli_masks)
            from .nn.modules.dropout import bernoulli_mask
            batch = getattr(self.mod, "upload_bernoulli_masks", None)
            if batch is None or not batch(self.feeds):
                for mask, keep in self.feeds:
                    self.mod.from_numpy(np.ascontiguousarray(bernoulli_mask(keep, mask.shape)), mask._handle)
        self.graph.replay()
        if self.sync is not None and not self.exchange_captured:
            for p, g in zip(opt.params, self.grads):
                p.grad = g
            self.sync()
            opt.apply()
        else:
            opt.note_replayed_apply()               # the captured apply() moved the device's scalar table on
        opt.t += 1
        return self.loss

    def destroy(self):
        self.graph.destroy()


class OverlappedCapturedTrainStep:
    """Data-parallel step with the gradient exchange overlapped with the backward pass.

    The backward sweep is captured as TWO graphs split at `model.backward_milestone` (for the CIFAR
    ResNet: the input of layer4 — behind it sit 2.4 M of the 2.8 M parameters, and their gradients
    are final first).  A replay is

        graph A (forward, loss, backward down to the milestone)
        pack + all-reduce of the late bucket      -> communication stream
        graph B (the rest of backward)            -> library stream, beside the all-reduce
        pack + all-reduce of the early bucket     -> communication stream
        wait, Adam

    so the 11 MB exchange of a step hides behind ~0.6 ms of backward instead of extending the step.
    No collective is inside a graph.  `late_bucket` / `early_bucket` are GradientBuckets over the two
    halves of optimizer.params (workloads.late_and_early_parameters)."""

    def __init__(self, model, criterion, optimizer, inputs, targets, late_bucket, early_bucket):
        mod = inputs.device.mod
        if not (getattr(mod, "fused", False) and hasattr(mod, "capture_begin")):
            raise RuntimeError("OverlappedCapturedTrainStep needs the hip device")
        self.mod, self.optimizer = mod, optimizer
        self.late, self.early = late_bucket, early_bucket
        optimizer.device_hyper = True
        optimizer.push_hyper()
        self.graph_a = None

        def split():
            self.graph_a = mod.capture_end()
            mod.capture_begin()

        mod.capture_begin()
        try:
            outputs = model(inputs)
            loss = criterion(outputs, targets)
            optimizer.zero_grad()
            node = getattr(model, "backward_milestone", None)
            if node is None or not node.requires_grad:
                raise RuntimeError("the model did not record a backward milestone")
            loss.backward(milestone=(node, split))
            if self.graph_a is None:
                raise RuntimeError("the backward sweep never reached the milestone")
            Graph.free_graph()
        except BaseException:
            try:
                mod.capture_end().destroy()
            except Exception:
                pass
            if self.graph_a is not None:
                self.graph_a.destroy()
            raise
        self.graph_b = mod.capture_end()
        self.loss, self.outputs = loss, outputs
        self._keep = (inputs, targets)
        self.late_grads = [p.grad for p in self.late.params]
        self.early_grads = [p.grad for p in self.early.params]
        if any(g is None for g in self.late_grads + self.early_grads):
            raise RuntimeError("a parameter received no gradient in the captured step")

    def replay(self):
        opt = self.optimizer
        opt.push_hyper()
        self.graph_a.replay()
        for p, g in zip(self.late.params, self.late_grads):
            p.grad = g
        self.late.sync(overlap=True)
        self.graph_b.replay()
        for p, g in zip(self.early.params, self.early_grads):
            p.grad = g
        self.early.sync(overlap=True)
        self.late.comm.wait()
        opt.apply()
        opt.t += 1
        return self.loss

    def destroy(self):
        self.graph_a.destroy()
        self.graph_b.destroy()
