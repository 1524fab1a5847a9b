"""Device-side input pipeline for the CIFAR training scripts.

The reference augments every batch on the host (test/ResNet_CIFAR10_cuda.py:129-148: reflect pad,
random crop, flip, optional random erasing, clip), one-hot encodes and smooths the labels
(:180-184) and then copies both synchronously (`Tensor(..., device='cuda')`).  Here the host only
draws the per-image random decisions — in the reference's NumPy RNG call order, so a seeded run
makes the same choices — and hands the raw batch, the labels and the decisions to a pinned,
double-buffered upload on a copy stream; crop / flip / erase / clip and the label smoothing are
kernels that write straight into the tensors the (captured) training step reads.

    pipe = DeviceBatchPipeline(dev, (256, 3, 32, 32), num_classes=10, eps=0.05)
    pipe.submit(x0, y0, epoch, num_epochs)                 # upload of batch 0 starts
    for k in range(steps):
        pipe.deliver(xt, yt)                               # augment batch k into the step's inputs
        if k + 1 < steps: pipe.submit(x[k + 1], ...)       # batch k+1 uploads during step k
        step.replay()
"""
import builtins

import numpy as np


def draw_decisions(bs, h, w, epoch, num_epochs, pad=4, rng=np.random):
    """The random choices of augment_batch in its own call order; int32 (bs, 8) rows
    {ys, xs, flip, erase_y, erase_x, erase_h, erase_w, 0}."""
    ys = rng.randint(0, 2 * pad + 1, size=bs)
    xs = rng.randint(0, 2 * pad + 1, size=bs)
    flip = rng.rand(bs) < 0.5
    dec = np.zeros((bs, 8), dtype=np.int32)
    dec[:, 0], dec[:, 1], dec[:, 2] = ys, xs, flip
    if epoch < num_epochs - 5 and rng.rand() < 0.2:
        eh = builtins.max(1, int(h * rng.uniform(0.1, 0.2)))
        ew = builtins.max(1, int(w * rng.uniform(0.1, 0.2)))
        dec[:, 3] = rng.randint(0, h - eh + 1, size=bs)
        dec[:, 4] = rng.randint(0, w - ew + 1, size=bs)
        dec[:, 5], dec[:, 6] = eh, ew
    return dec


class _Window:
    """Non-owning window into a device array (duck type of a device handle: .ptr, .size)."""
    __slots__ = ("ptr", "size", "_owner")

    def __init__(self, owner, offset, size):
        self.ptr, self.size, self._owner = owner.ptr + 4 * int(offset), int(size), owner


class DeviceBatchPipeline:
    """`augment=False`: the scripts without augmentation (test/CNN_CIFAR10_cuda.py:128-136, test/MLP_MNIST_cuda.py)
    -- the raw batch is copied from the staging slot into the step's input (no random draws, no clip); labels are
    still turned into (smoothed) one-hot rows on the device.  `batch_shape` may then have any rank (MLP: (bs, 784))."""

    def __init__(self, device, batch_shape, num_classes=10, eps=0.05, pad=4, augment=True):
        mod = device.mod
        if not hasattr(mod, "upload_async"):
            raise RuntimeError("DeviceBatchPipeline needs the hip device")
        self.mod, self.shape, self.K, self.eps, self.pad = mod, tuple(int(s) for s in batch_shape), int(num_classes), eps, pad
        self.augment = bool(augment)
        if self.augment and len(self.shape) != 4:
            raise ValueError("augmentation needs (batch, channels, height, width) batches")
        n = int(np.prod(self.shape))
        bs = self.shape[0]
        # one staging array per slot: [raw batch | labels (int32) | decisions (int32 x 8)]
        self._off_lab, self._off_dec = n, n + bs
        self._stage = [mod.Array(n + bs + 8 * bs) for _ in range(2)]
        self._next_submit, self._next_deliver = 0, 0

    def submit(self, inputs, labels, epoch=0, num_epochs=1, rng=np.random):
        """Start the upload of one raw batch (float32 NCHW) and its integer labels."""
        bs = self.shape[0]
        assert tuple(inputs.shape) == self.shape
        if self.augment:
            dec = draw_decisions(bs, self.shape[2], self.shape[3], epoch, num_epochs, self.pad, rng)
        else:
            dec = np.zeros((bs, 8), dtype=np.int32)
        slot = self._next_submit & 1
        packed = np.empty(self._stage[slot].size, dtype=np.float32)
        packed[:self._off_lab] = np.asarray(inputs, dtype=np.float32).reshape(-1)
        packed[self._off_lab:self._off_dec] = np.asarray(labels, dtype=np.int32).reshape(-1).view(np.float32)
        packed[self._off_dec:] = dec.reshape(-1).view(np.float32)
        self.mod.upload_async(packed, self._stage[slot], slot)
        self._next_submit += 1
        return dec

    def deliver(self, x_tensor, y_tensor):
        """Augment the oldest submitted batch into x_tensor / y_tensor (device tensors of the step)."""
        assert self._next_deliver < self._next_submit, "deliver() without a submitted batch"
        slot = self._next_deliver & 1
        bs = self.shape[0]
        st = self._stage[slot]
        self.mod.upload_wait(slot)
        if self.augment:
            _, c, h, w = self.shape
            self.mod.augment_batch(st, x_tensor.data._handle, st, bs, c, h, w, self.pad, 0, self._off_dec)
        else:
            self.mod.copy(_Window(st, 0, self._off_lab), _Window(x_tensor.data._handle, 0, self._off_lab))
        self.mod.onehot_smooth(st, y_tensor.data._handle, bs, self.K, self.eps, self._off_lab)
        if hasattr(self.mod, "upload_release"):
            self.mod.upload_release(slot)       # (the staging slot is free for the batch after next from HERE on the stream)
        self._next_deliver += 1
