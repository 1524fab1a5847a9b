"""The per-epoch loops of the reference's training scripts, driven from the device side.

Reference (test/ResNet_CIFAR10_cuda.py:173-236; the CNN and MLP scripts are the same loop without augmentation):

    for batch_idx, (inputs, labels) in enumerate(loader):          # data_loader(x, y, bs, shuffle=True)
        inputs = augment_batch(inputs, epoch, num_epochs)           # host NumPy, global generator
        labels_onehot = one_hot(labels) * (1 - eps) + eps / K
        inputs, labels_onehot = Tensor(inputs, device), Tensor(labels_onehot, device)   # synchronous H2D
        outputs = model(inputs); loss = criterion(outputs, labels_onehot)
        optimizer.zero_grad(); loss.backward(); optimizer.step()
        loss_value = loss.data.numpy().item()                       # D2H + sync every batch

Here (hip device only -- like every product path, `Device('hip')` raises when the extension or the GPU is missing):
  * batches come from `utils.data.DataLoader` (same sampler / batch order as the reference's);
  * batch k + 1 is uploaded through `DeviceBatchPipeline` (pinned double buffer, copy stream) while step k runs;
    crop / flip / erase / clip and the label smoothing are kernels that write into the tensors the step reads;
  * full-size batches replay ONE captured hipGraph (`CapturedTrainStep`); the ragged last batch of an epoch (50 000
    images = 195 x 256 + 80) runs the same step eagerly on tensors of its own size (`tail="eager"`) or through a second
    captured graph of that size (`tail="capture"`); both read and update the same parameter, optimizer-state and
    BatchNorm running-statistics buffers in place, so the paths interleave freely;
  * losses stay on the device: each step's scalar is copied into a slot of a device array and the epoch's losses are
    read back once (`run_epoch` returns them), instead of one blocking read per batch.
Host random numbers are drawn in the reference's order: [permutation at the first batch], then per batch the
augmentation decisions, then whatever the forward pass draws (Dropout masks) -- the look-ahead upload of batch k + 1
draws ITS decisions after step k has drawn its masks.
"""
import numpy as np

from .. import workloads
from ..autograd import no_grad
from ..tensor import Graph, Tensor
from .data_pipeline import DeviceBatchPipeline, _Window


class EpochTrainer:
    def __init__(self, model, criterion, optimizer, device, num_classes=10, eps=0.05, augment=True, capture=True,
                 tail="eager", sync=None, max_losses=4096):
        mod = device.mod
        if not (getattr(mod, "fused", False) and hasattr(mod, "upload_async")):
            raise RuntimeError("EpochTrainer needs the hip device")
        if tail not in ("eager", "capture"):
            raise ValueError("tail is 'eager' or 'capture'")
        self.model, self.criterion, self.optimizer, self.device, self.mod = model, criterion, optimizer, device, mod
        self.K, self.eps, self.augment, self.capture, self.tail, self.sync = int(num_classes), eps, augment, capture, tail, sync
        self._slots = {}             # batch shape -> (pipeline, x tensor, y tensor)
        self._captured = {}          # batch shape -> CapturedTrainStep
        self._losses = mod.Array(int(max_losses))
        self._main_shape = None

    # ---- per-shape resources ----------------------------------------------------------------------------
    def _slot(self, shape):
        got = self._slots.get(shape)
        if got is None:
            pipe = DeviceBatchPipeline(self.device, shape, num_classes=self.K, eps=self.eps, augment=self.augment)
            xt = Tensor(np.zeros(shape, np.float32), device=self.device)
            yt = Tensor(np.zeros((shape[0], self.K), np.float32), device=self.device)
            got = self._slots[shape] = (pipe, xt, yt)
        return got

    def _step(self, shape, xt, yt):
        """One training step on the tensors of `shape`; returns the device tensor holding the loss."""
        use_graph = self.capture and (shape == self._main_shape or self.tail == "capture")
        if use_graph:
            cs = self._captured.get(shape)
            if cs is None:
                from ..graph_step import CapturedTrainStep
                # (a capture records the step without running it and draws no host random numbers: Dropout masks
                # are graph inputs that every replay draws, nn/modules/dropout.py)
                cs = self._captured[shape] = CapturedTrainStep(self.model, self.criterion, self.optimizer, xt, yt,
                                                               sync=self.sync)
            return cs.replay()
        loss, _ = workloads.train_step(self.model, self.criterion, self.optimizer, xt, yt, sync=self.sync)
        Graph.free_graph()
        return loss

    # ---- one epoch -----------------------------------------------------------------------------------------
    def run_epoch(self, loader, epoch=0, num_epochs=1, max_batches=None):
        """Train on every batch `loader` yields (at most `max_batches`, as the scripts' `max_train_batches`);
        returns the per-batch losses (float32 array), read back once at the end."""
        self.model.train()
        it = iter(loader)

        def pull(k):
            if max_batches is not None and k >= max_batches:
                return None
            try:
                x, y = next(it)
            except StopIteration:
                return None
            x = np.asarray(x, dtype=np.float32)
            shape = tuple(int(s) for s in x.shape)
            pipe, xt, yt = self._slot(shape)
            pipe.submit(x, np.asarray(y).reshape(-1), epoch, num_epochs)
            return shape

        n = 0
        pending = pull(0)
        if pending is not None and self._main_shape is None:
            self._main_shape = pending          # the first batch of the first epoch has the loader's batch size
        while pending is not None:
            if n >= self._losses.size:
                raise RuntimeError(f"more than {self._losses.size} batches in an epoch: raise max_losses")
            shape = pending
            pipe, xt, yt = self._slot(shape)
            pipe.deliver(xt, yt)
            loss = self._step(shape, xt, yt)
            self.mod.copy(_Window(loss.data.compact()._handle, 0, 1), _Window(self._losses, n, 1))
            n += 1
            pending = pull(n)                   # upload of the next batch overlaps the step just queued
        out = self.mod.to_numpy(self._losses, (max(n, 1),), (1,), 0)[:n].copy()
        return out

    def evaluate(self, loader, max_batches=None):
        """The scripts' test pass (test/ResNet_CIFAR10_cuda.py:205-234): eval mode, no tape, arg-max accuracy in
        percent over the batches of `loader` (labels are class indices)."""
        self.model.eval()
        correct = total = 0
        try:
            with no_grad():
                for k, (x, y) in enumerate(loader):
                    if max_batches is not None and k >= max_batches:
                        break
                    out = self.model(Tensor(np.asarray(x, dtype=np.float32), device=self.device))
                    pred = np.argmax(out.numpy(), 1).reshape(-1)
                    correct += int(np.sum(pred == np.asarray(y).reshape(-1)))
                    total += int(pred.shape[0])
                    Graph.free_graph()
        finally:
            self.model.train()
        return 100.0 * correct / max(total, 1)

    def destroy(self):
        for cs in self._captured.values():
            cs.destroy()
        self._captured.clear()


def train_epoch(trainer, loader, epoch, num_epochs, scheduler=None, max_batches=None):
    """One epoch of the scripts' outer loop: train, then step the learning-rate schedule (:235)."""
    losses = trainer.run_epoch(loader, epoch, num_epochs, max_batches)
    if scheduler is not None:
        scheduler.step()
    return losses
