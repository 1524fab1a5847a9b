"""TEST INFRASTRUCTURE: a gloo communicator on host memory for the world_size-2 CPU tests with the NumPy oracle
device (its handles are numpy arrays, which torch wraps without copying).  The product's communicator is
deepflows_amd.parallel.RcclCommunicator; nothing in the package imports torch."""
import os

import numpy as np

from deepflows_amd.parallel import Communicator


class GlooCommunicator(Communicator):
    exchanges = True

    def __init__(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", init_method="env://")
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allreduce_sum(self, flat):
        import torch
        h = flat._handle
        if isinstance(h, np.ndarray):
            self._dist.all_reduce(torch.from_numpy(h))
        else:                                   # a device array: stage through the host
            a = flat.numpy()
            self._dist.all_reduce(torch.from_numpy(a))
            flat.device.from_numpy(a, h)
        return flat

    def barrier(self):
        self._dist.barrier()

    def max_float(self, x):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())
