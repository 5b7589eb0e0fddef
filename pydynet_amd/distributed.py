"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no parallelism of any kind; this is new functionality whose
contract is: an N-rank step on N shards of a batch == a 1-rank step on the concatenated batch
(loss is a mean over tokens, so averaging shard gradients gives the global gradient).

Design for MI355X (8 GPUs fully connected, 7 xGMI links x ~153 GB/s each):
  * all trainable gradients live in ONE flat fp32 buffer laid out in reverse registration order
    (lm_head first, tok_embedding last) -- the order backward finalises them;
  * the buffer is cut into buckets (default 12 MB: a ring all-reduce is per-link bound, so few
    large messages beat many small ones, but a bucket can only start once its LAST gradient exists --
    12 MB lets the transformer blocks' 24 MB go out in two pieces during backward and leaves only the
    embedding table, whose gradient is the final one, exposed); each parameter's `.grad` is a view
    into its bucket;
  * the tape engine fires a grad-ready hook when a leaf has received its last contribution;
    when every parameter of bucket k is ready (and buckets < k are already in flight) the
    bucket's `all_reduce(SUM)` is issued asynchronously -- RCCL runs it on its own stream, so
    layer k's communication overlaps layer k-1's backward kernels;
  * the 1/N average is not a separate pass: it is folded into the Adam kernel (`grad_scale`).
`torch.distributed` (backend "nccl" == RCCL on ROCm, "gloo" for CPU tests) is the plumbing.
"""
from __future__ import annotations

import os

import numpy as np

from .core.tensor import Tensor


def _dist():
    import torch.distributed as dist
    return dist


def init_process_group(backend=None, device_index=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world)."""
    import torch
    dist = _dist()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device_index is not None:
            kw["device_id"] = torch.device(f"cuda:{device_index}")
        dist.init_process_group(backend=backend, **kw)
    return rank, world


def shard_batch(global_batch: int, rank: int, world: int):
    """Rows [lo, hi) of a global batch owned by `rank` (equal shards required)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


class DataParallel:
    """Wraps a Module: broadcasts parameters from rank 0, flattens gradients into buckets and
    overlaps their all-reduce with backward.  Usage:

        dp = DataParallel(model, optimizer)          # after model.to(device) and Adam(...)
        optimizer.zero_grad(); loss = model.loss(...); loss.backward(); dp.finish(); optimizer.step()
    """

    def __init__(self, module, optimizer=None, bucket_mb: float = 12.0, process_group=None,
                 broadcast_parameters=True, always_reduce=False):
        dist = _dist()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # `always_reduce` runs the collectives even on a single rank (used to exercise the RCCL path
        # on a one-GPU box); normally a lone rank skips them
        self._comm = dist.is_initialized() and (self.world > 1 or always_reduce)
        self.params = [p for p in module.parameters()][::-1]        # reverse registration order
        if not self.params:
            raise ValueError("DataParallel: module has no trainable parameters")
        self.device = self.params[0].device
        self._flatten(bucket_mb)
        for i, p in enumerate(self.params):
            p._grad_hook = self._make_hook(i)
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
            optimizer._table_key = None               # grads moved: rebuild the chunk table
            if [id(p) for p in optimizer.params][::-1] == [id(p) for p in self.params]:
                optimizer._flat_grad, optimizer._flat_offsets = self.flat, self.offsets[::-1]
                optimizer._flat_views = [p.grad for p in optimizer.params]
        if broadcast_parameters and self._comm:
            self.broadcast_parameters()
        self._reset()

    # -- flat gradient storage ----------------------------------------------------------------
    def _flatten(self, bucket_mb):
        from .optim.flat import flatten_gradients
        sizes = [p.size for p in self.params]
        self.flat, offs = flatten_gradients(self.params)
        self.offsets = offs
        cap = max(int(bucket_mb * (1 << 20) / 4), 1)
        self.buckets, start, first = [], 0, 0        # (elem_lo, elem_hi, param_lo, param_hi)
        for i, (off, n) in enumerate(zip(offs, sizes)):
            end = off + (n + 3) // 4 * 4
            if end - start >= cap or i == len(sizes) - 1:
                self.buckets.append((start, end, first, i + 1))
                start, first = end, i + 1
        self.param_bucket = {}
        for b, (_, _, lo, hi) in enumerate(self.buckets):
            for i in range(lo, hi):
                self.param_bucket[i] = b

    def _torch_view(self, lo, hi):
        seg = self.flat[lo:hi]
        if isinstance(seg, np.ndarray):
            import torch
            return torch.from_numpy(seg)
        return seg.as_torch()

    # -- hooks ------------------------------------------------------------------------------------
    def _reset(self):
        self._ready = [0] * len(self.buckets)
        self._next = 0
        self._works = []

    def _make_hook(self, index):
        def hook(_param):
            b = self.param_bucket[index]
            self._ready[b] += 1
            self._launch_ready()
        return hook

    def _launch_ready(self):
        if not self._comm:
            return
        dist = _dist()
        while self._next < len(self.buckets):
            lo, hi, plo, phi = self.buckets[self._next]
            if self._ready[self._next] < phi - plo:
                break
            self._works.append(dist.all_reduce(self._torch_view(lo, hi), op=dist.ReduceOp.SUM,
                                               group=self.group, async_op=True))
            self._next += 1

    def finish(self):
        """Call after backward(): issues any bucket not yet launched (parameters that received no
        gradient this step) and waits for all reductions.  Gradients then hold the SUM over ranks;
        the optimizer divides by the world size through `grad_scale`."""
        if self._comm:
            dist = _dist()
            while self._next < len(self.buckets):
                lo, hi, _, _ = self.buckets[self._next]
                self._works.append(dist.all_reduce(self._torch_view(lo, hi), op=dist.ReduceOp.SUM,
                                                   group=self.group, async_op=True))
                self._next += 1
            for w in self._works:
                w.wait()
        self._reset()

    def zero_grad(self):
        with self.device:
            self.flat[...] = 0.0

    def broadcast_parameters(self, src=0):
        dist = _dist()
        for p in self.module._parameters.values():
            if isinstance(p.data, np.ndarray):
                import torch
                t = torch.from_numpy(p.data)
            else:
                t = p.data.as_torch() if p.data.is_contiguous() else None
                if t is None:
                    raise ValueError("DataParallel: non-contiguous parameter")
            dist.broadcast(t, src=src, group=self.group)

    def __call__(self, *a, **kw):
        return self.module(*a, **kw)
