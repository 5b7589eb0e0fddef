"""Test-only communicator: torch.distributed (gloo) moving NumPy buffers, registered with the product through
`pydynet_amd.distributed.register_backend("gloo", ...)`.  It exists so that the N > 1 logic of DataParallel
(buckets, grad-ready hooks, parameter broadcast, the embedding owner vote) runs in the GPU-less container; it is
never used with HIP arrays and is not part of the product."""
import numpy as np

from pydynet_amd.distributed import SUM, MAX, register_backend  # noqa: F401


class GlooComm:
    """torch.distributed (gloo) moving NumPy buffers: the communicator of the "cpu" device.  It exists
    so that the N > 1 logic runs in GPU-less tests; it is never used with HIP arrays."""

    backend = "gloo"

    def __init__(self, rank: int, world: int):
        import torch.distributed as dist
        self._dist = dist
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._works = []

    def _tensor(self, arr):
        import torch
        assert isinstance(arr, np.ndarray) and arr.flags.c_contiguous
        return torch.from_numpy(arr)

    def all_reduce(self, arr, op=SUM):
        dist = self._dist
        self._works.append(dist.all_reduce(self._tensor(arr), async_op=True,
                                           op=dist.ReduceOp.SUM if op == SUM else dist.ReduceOp.MAX))

    def broadcast(self, arr, root=0):
        self._works.append(self._dist.broadcast(self._tensor(arr), src=root, async_op=True))

    def all_gather(self, send, recv):
        import torch
        parts = list(torch.from_numpy(recv.reshape(self.world, -1)).unbind(0))
        self._works.append(self._dist.all_gather(parts, self._tensor(send).reshape(-1), async_op=True))

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []

    def barrier(self):
        self.wait()
        self._dist.barrier()

    def all_reduce_scalar(self, value: float, op=MAX) -> float:
        a = np.array([value], np.float64)
        self.all_reduce(a, op)
        self.wait()
        return float(a[0])

    def destroy(self):
        self.wait()
        if self._dist.is_initialized():
            self._dist.destroy_process_group()


def install():
    register_backend("gloo", GlooComm)
