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
  * round 4: the backward kernels WRITE the gradients straight into the bucket -- every registered parameter carries
    its slice of the bucket as `grad_slot`, and the fused nodes take it as the output array of the kernel that forms
    the parameter's gradient (nn/functional.py: _grad_out) -- so the exchange step is the all-reduce alone: no pack
    launch, no second copy of the 11 MB.  A parameter whose gradient did not end up in its slot (it received two
    contributions, or came from a node that allocates its own output) is copied in as before;
  * BatchNorm statistics and dropout masks stay rank-local (no SyncBN), the loss is the mean
    over the local batch.
world == 1 degenerates to "no bucket, no collective" so the single-GPU path is bit-identical
with and without this wrapper.
"""
import os

import numpy as np

from .backend_selection import BackendTensor


# gradients are formed in the bucket itself (DF_GRAD_IN_BUCKET=0: packed by a launch of their own, as in rounds 1-3)
IN_PLACE = os.environ.get("DF_GRAD_IN_BUCKET", "1") != "0"


def _same_array(a, b):
    """two compact BackendTensors over the same device memory"""
    ha, hb = a._handle, b._handle
    if isinstance(ha, np.ndarray) or isinstance(hb, np.ndarray):
        return isinstance(ha, np.ndarray) and isinstance(hb, np.ndarray) and ha.shape == hb.shape and \
            ha.__array_interface__["data"][0] == hb.__array_interface__["data"][0]
    return getattr(ha, "ptr", None) is not None and ha.ptr == hb.ptr and ha.size == hb.size


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
    exchanges = False      # True: gradients go through a flat bucket and allreduce_sum even when world == 1

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


class RcclCommunicator(Communicator):
    """RCCL communicator owned by libdfhip (dfhip_comm_init / dfhip_allreduce_sum).  Rank 0 draws the 128-byte
    ncclUniqueId and serves it over a TCP socket (rendezvous.py, standard library); barrier and max-over-ranks are
    one-element collectives on the communicator itself.  No host-side process group, no torch."""
    exchanges = True

    def __init__(self, hip_mod, rank, world):
        from . import rendezvous
        self._mod = hip_mod
        self.rank, self.world = int(rank), int(world)
        uid = hip_mod.comm_unique_id() if self.rank == 0 else b""
        uid = rendezvous.broadcast_from_rank0(uid, self.rank, self.world)
        if len(uid) != 128:
            raise RuntimeError(f"rendezvous returned {len(uid)} bytes instead of the 128-byte ncclUniqueId")
        hip_mod.comm_init(self.rank, self.world, uid)

    def allreduce_sum(self, flat, overlap=False):
        if overlap:
            self._mod.allreduce_sum_async(flat._handle)   # communication stream, see wait()
        else:
            self._mod.allreduce_sum(flat._handle)
        return flat

    def wait(self):
        """The library stream waits for every overlapped all-reduce issued so far."""
        self._mod.comm_wait()

    def barrier(self):
        self._mod.comm_allreduce_max(0.0)

    def max_float(self, x):
        return self._mod.comm_allreduce_max(x)

    def close(self):
        self._mod.comm_destroy()


def seed_rank_local(seed, rank):
    """Rank-local host random numbers of a data-parallel job ("BN running stats and Dropout masks stay rank-local
    (seed + rank)").  Called AFTER the replicas have been built from the common seed: rank 0 keeps the stream where the
    initialisation left it -- a job of one rank is the single-GPU run bit for bit --, rank r > 0 continues from
    np.random.seed(seed + r), so no two ranks draw the same Dropout masks (`nn/modules/dropout.py` reads the global generator,
    as `DeepFlows/nn/modules/dropout.py:17` does)."""
    rank = int(rank)
    if rank > 0:
        np.random.seed((int(seed) + rank) & 0xFFFFFFFF)


def init_communicator(device):
    """Build the communicator for the launch environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as a
    launcher such as torch.distributed.run sets them; the launcher only spawns the processes).  world == 1 needs no
    communicator.  Only the hip device has collectives: anything else raises (the CPU tests bring their own gloo
    communicator, tests/gloo_communicator.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    force = os.environ.get("DF_FORCE_COMM", "") == "1"       # tests: exercise the collective with one rank
    if world <= 1 and not force:
        return LocalCommunicator()
    if getattr(device.mod, "fused", False) and hasattr(device.mod, "comm_init"):
        return RcclCommunicator(device.mod, rank, max(1, world))
    raise RuntimeError("init_communicator: the device has no RCCL entry points (multi-GPU runs need the hip device)")


class GradientBucket:
    """Flat gradient bucket over a parameter list + the exchange step."""

    def __init__(self, params, comm):
        self.params = [p for p in params]
        self.comm = comm
        self.sizes = [int(p.size) for p in self.params]
        self.total = sum(self.sizes)
        self.flat = None
        self.slots = []
        if (comm.world > 1 or comm.exchanges) and self.params:
            dev = self.params[0].device
            self.flat = BackendTensor.make((self.total,), device=dev)
            self.flat.fill(0.0)
            if IN_PLACE:
                off = 0
                for p, n in zip(self.params, self.sizes):
                    slot = self._view(off, p)
                    self.slots.append(slot)
                    p.grad_slot = slot            # (the kernels that form p's gradient write it here)
                    off += n
        self.packed_last = None                   # how many gradients the last sync() had to copy into the bucket

    def release(self):
        """the parameters stop pointing at this bucket (another bucket over the same parameters takes over)"""
        for p in self.params:
            if getattr(p, "grad_slot", None) is not None:
                p.grad_slot = None
        self.slots = []

    def attach(self):
        """the parameters point at THIS bucket's slices again (after another bucket over the same parameters was built and
        abandoned: every constructor re-assigns p.grad_slot)"""
        if self.flat is not None and IN_PLACE:
            self.slots = []
            off = 0
            for p, n in zip(self.params, self.sizes):
                slot = self._view(off, p)
                self.slots.append(slot)
                p.grad_slot = slot
                off += n

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
        grads, outside = [], 0
        for i, p in enumerate(self.params):
            if p.grad is None:
                raise RuntimeError("GradientBucket.sync: a parameter has no gradient")
            g = p.grad.compact()
            grads.append(g)
            if not (self.slots and _same_array(g, self.slots[i])):
                outside += 1
        self.packed_last = outside
        if outside == 0:
            pass                                  # every gradient was written in place: nothing to pack
        elif outside < len(grads) and hasattr(dev.mod, "copy") and not isinstance(self.flat._handle, np.ndarray):
            for i, g in enumerate(grads):
                if not _same_array(g, self.slots[i]):
                    dev.mod.copy(g._handle, self.slots[i]._handle)
        else:
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
