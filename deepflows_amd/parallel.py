"""Data-parallel training step: one process per GPU, a full model replica per rank, ONE
exchange step per batch — the all-reduce of the gradient bucket over RCCL / xGMI.

The reference has no distributed code at all (no NCCL, no collective); this layer
is the new functionality BASELINE's north star asks for (§8e).  Design for MI355X:
  * all registered parameter gradients are packed by one multi-tensor kernel into ONE flat fp32
    bucket (ResNet-18 CIFAR variant: 2 797 610 elements = 11.2 MB).  xGMI is point-to-point
    (7 links x ~153 GB/s per GPU): at this size the collective is latency-bound, so one large
    message beats many small ones;
  * `ncclAllReduce(sum)` runs in place on the library stream, ordered after the backward
    kernels that produced the gradients and before the optimizer kernel that consumes them —
    no host synchronisation anywhere in the step;
  * the optimizer reads the gradients straight out of the bucket (views, no unpack) and applies
    the 1/world average inside its fused kernel (`grad_scale`);
  * BatchNorm statistics and dropout masks stay rank-local (no SyncBN), the loss is the mean
    over the local batch.
world == 1 degenerates to "no bucket, no collective" so the single-GPU path is bit-identical
with and without this wrapper.
"""
import os

import numpy as np

from .backend_selection import BackendTensor


class _ArrayView:
    """Non-owning window into a flat device array (same duck type as a device handle)."""
    __slots__ = ("ptr", "size", "_owner")

    def __init__(self, owner, offset, size):
        self.ptr = owner.ptr + 4 * offset
        self.size = size
        self._owner = owner


class Communicator:
    rank = 0
    world = 1

    def allreduce_sum(self, flat):          # flat: BackendTensor, 1-D compact
        raise NotImplementedError

    def barrier(self):
        pass

    def max_float(self, x):
        return x

    def wait(self):
        pass

    def close(self):
        pass


class LocalCommunicator(Communicator):
    def allreduce_sum(self, flat):
        return flat


class TorchHostCommunicator(Communicator):
    """gloo on host memory — used by the world_size-2 CPU tests with the NumPy oracle device
    (its handles are numpy arrays, which torch wraps without copying)."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allreduce_sum(self, flat):
        import torch
        h = flat._handle
        if isinstance(h, np.ndarray):
            self._dist.all_reduce(torch.from_numpy(h))
        else:                                   # a device array: stage through the host
            a = flat.numpy()
            t = torch.from_numpy(a)
            self._dist.all_reduce(t)
            flat.device.from_numpy(a, h)
        return flat

    def barrier(self):
        self._dist.barrier()

    def max_float(self, x):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())


class RcclCommunicator(TorchHostCommunicator):
    """RCCL communicator owned by libdfhip (dfhip_comm_init / dfhip_allreduce_sum).  The
    128-byte ncclUniqueId is broadcast through the already-initialised torch.distributed
    (gloo) group — rendezvous plumbing only; the data path never touches torch."""

    def __init__(self, hip_mod):
        super().__init__()
        import torch
        self._mod = hip_mod
        uid = hip_mod.comm_unique_id() if self.rank == 0 else bytes(128)
        t = torch.tensor(list(uid), dtype=torch.uint8)
        self._dist.broadcast(t, src=0)
        hip_mod.comm_init(self.rank, self.world, bytes(t.tolist()))

    def allreduce_sum(self, flat, overlap=False):
        if overlap:
            self._mod.allreduce_sum_async(flat._handle)   # communication stream, see wait()
        else:
            self._mod.allreduce_sum(flat._handle)
        return flat

    def wait(self):
        """The library stream waits for every overlapped all-reduce issued so far."""
        self._mod.comm_wait()

    def close(self):
        self._mod.comm_destroy()


def init_communicator(device):
    """Build the communicator for the launch environment (RANK / WORLD_SIZE / MASTER_* as set
    by torch.distributed.run).  world == 1 needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("DF_FORCE_COMM", "") == "1"       # tests: exercise the collective with one rank
    if world <= 1 and not force:
        return LocalCommunicator()
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", init_method="env://")
    if getattr(device.mod, "fused", False) and hasattr(device.mod, "comm_init"):
        return RcclCommunicator(device.mod)
    return TorchHostCommunicator()


class GradientBucket:
    """Flat gradient bucket over a parameter list + the exchange step."""

    def __init__(self, params, comm):
        self.params = [p for p in params]
        self.comm = comm
        self.sizes = [int(p.size) for p in self.params]
        self.total = sum(self.sizes)
        self.flat = None
        if (comm.world > 1 or isinstance(comm, TorchHostCommunicator)) and self.params:
            dev = self.params[0].device
            self.flat = BackendTensor.make((self.total,), device=dev)
            self.flat.fill(0.0)

    def _view(self, offset, p):
        h = self.flat._handle
        if isinstance(h, np.ndarray):
            handle = h[offset:offset + p.size]
        else:
            handle = _ArrayView(h, offset, int(p.size))
        return BackendTensor.make(p.shape, device=p.device, handle=handle)

    def sync(self, overlap=False):
        """pack -> all-reduce(sum) -> point every p.grad at its slice of the reduced bucket.
        The optimizer must then be stepped with grad_scale = 1/world.  overlap=True: the all-reduce
        runs on the communication stream beside whatever is enqueued next; call comm.wait() before
        the optimizer."""
        if self.flat is None:
            return
        dev = self.params[0].device
        grads = []
        for p in self.params:
            if p.grad is None:
                raise RuntimeError("GradientBucket.sync: a parameter has no gradient")
            grads.append(p.grad.compact())
        pack = getattr(dev.mod, "pack", None)
        if pack is not None:
            pack([g._handle for g in grads], self.flat._handle)
        else:
            off = 0
            for g, n in zip(grads, self.sizes):
                self.flat._handle[off:off + n] = g._handle
                off += n
        if overlap and isinstance(self.comm, RcclCommunicator):
            self.comm.allreduce_sum(self.flat, overlap=True)
        else:
            self.comm.allreduce_sum(self.flat)
        off = 0
        for p, n in zip(self.params, self.sizes):
            p.grad = self._view(off, p)
            off += n
