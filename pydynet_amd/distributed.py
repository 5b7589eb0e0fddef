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
    bucket's all-reduce(SUM) is enqueued on a dedicated COMMUNICATION stream behind an
    event recorded on the compute stream -- layer k's reduction overlaps layer k-1's backward
    kernels; `finish()` makes the compute stream wait for the last bucket (no host sync);
  * the 1/N average is not a separate pass: every optimizer applies `grad_scale` (Adam folds it into
    its kernel);
  * the embedding gradient keeps the reference's scatter-ASSIGN semantics across ranks: an
    all-reduce(MAX) over a (V,) "owner" vector picks, for every token id, the highest rank that saw
    it (= the last occurrence in the concatenated batch) and only that rank contributes the row.
Collectives: RCCL through the C ABI (`pdn_comm_*`, include/pdn_hip.h) -- no PyTorch involved.  (The GPU-less
tests plug a host communicator in through `register_backend`; it lives under tests/.)
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import rendezvous

SUM, MAX = 0, 1
_group = None


class RcclComm:
    """One RCCL rank bound to the current GPU: collectives run on a communication stream of their
    own, ordered against the compute stream with events (never with host synchronisation)."""

    backend = "rccl"

    def __init__(self, rank: int, world: int, device_index: int = 0):
        from . import hipnp, _lib
        self.rank, self.world = int(rank), int(world)
        self._hp, self._L = hipnp, _lib.lib()
        L = self._L
        hipnp.set_device(device_index)
        self.device_index = device_index
        # CU budget of the collectives: an RCCL channel is one workgroup, and the GEMMs of the training step keep ONE
        # 8-wave workgroup on every CU -- a channel's workgroup co-resides with it and takes issue slots from it.  The
        # gradient exchange of a step (97.8 MB, ring all-reduce: 2 (N - 1) / N x that per GPU = 171 MB at N = 8) needs
        # ~0.5 ms of the seven xGMI links spread over a 57 ms backward, so a handful of channels is plenty: 8 (one per
        # link + one) unless the launcher says otherwise (NCCL_MAX_NCHANNELS / PDN_RCCL_MAX_CHANNELS; 0 = RCCL's default).
        # The variable is read by RCCL at ITS first initialisation in the process (a communicator created earlier by another
        # package has already fixed it: logged below) and is put back right after ncclCommInitRank, so child processes and
        # other NCCL users of this process do not inherit the cap.
        ch = os.environ.get("PDN_RCCL_MAX_CHANNELS", "8")
        self._nch_prev, self._nch_set = os.environ.get("NCCL_MAX_NCHANNELS"), False
        if ch != "0" and self._nch_prev is None:
            os.environ["NCCL_MAX_NCHANNELS"] = ch
            self._nch_set = True
        # what this communicator was initialised under (bench.py's `comm` block quotes it): None = RCCL's default
        self.channels_cap = os.environ.get("NCCL_MAX_NCHANNELS")
        if self.rank == 0 and os.environ.get("PDN_DP_QUIET") != "1":
            import sys
            print(f"[pydynet_amd.distributed] RCCL channels: NCCL_MAX_NCHANNELS={os.environ.get('NCCL_MAX_NCHANNELS', 'unset (RCCL default)')}"
                  f"{' (set here; PDN_RCCL_MAX_CHANNELS=0 leaves the default)' if self._nch_set else ''}", file=sys.stderr)
        uid = None
        if self.rank == 0:
            buf = ctypes.create_string_buffer(128)
            L.call("pdn_comm_unique_id", buf)
            uid = buf.raw
        uid = rendezvous.broadcast_bytes(uid, self.rank, self.world)
        h = ctypes.c_void_p()
        L.call("pdn_comm_init", ctypes.byref(h), self.rank, self.world, uid)
        self._comm = h.value
        if self._nch_set:                               # the cap was for THIS communicator's initialisation only
            os.environ.pop("NCCL_MAX_NCHANNELS", None)
        s = ctypes.c_void_p()
        # NORMAL priority on purpose: while a high-priority queue holds a barrier packet waiting for an
        # event, every kernel of the compute queue runs 10-60 us longer (one-GPU probe, B = 256:
        # 61.5 -> 68.8 ms per step with no collective issued at all; tools/dp_overhead_probe.py)
        L.call("pdn_stream_create", ctypes.byref(s), int(os.environ.get("PDN_COMM_HIGH_PRIORITY", "0")))
        self._stream = s.value
        self._events = []           # recycled event pairs
        self._pending = None        # last event recorded on the communication stream

    # -- event plumbing ---------------------------------------------------------------------------
    def _event(self):
        if self._events:
            return self._events.pop()
        e = ctypes.c_void_p()
        self._L.call("pdn_event_create", ctypes.byref(e), 0)
        return e.value

    def _after_compute(self):
        """Communication stream waits for everything enqueued on the compute stream so far."""
        e = self._event()
        self._L.call("pdn_event_record", e, self._hp.stream())
        self._L.call("pdn_stream_wait_event", self._stream, e)
        self._events.append(e)          # an event may be re-recorded once the wait is enqueued

    def _mark(self):
        if self._pending is None:
            self._pending = self._event()
        self._L.call("pdn_event_record", self._pending, self._stream)

    # -- collectives (asynchronous) -----------------------------------------------------------------
    def all_reduce(self, arr, op=SUM):
        """In place on a contiguous float32 device array."""
        assert arr.dtype == np.float32 and arr.is_contiguous()
        self._after_compute()
        self._L.call("pdn_comm_allreduce_f32", self._comm, arr._ptr, arr.size, op, self._stream)
        self._mark()

    def broadcast(self, arr, root=0):
        assert arr.is_contiguous()
        self._after_compute()
        self._L.call("pdn_comm_broadcast", self._comm, arr._ptr, arr.nbytes, root, self._stream)
        self._mark()

    def all_gather(self, send, recv):
        assert send.is_contiguous() and recv.is_contiguous() and recv.nbytes == send.nbytes * self.world
        self._after_compute()
        self._L.call("pdn_comm_allgather", self._comm, send._ptr, recv._ptr, send.nbytes, self._stream)
        self._mark()

    def wait(self):
        """Compute stream waits for every collective issued so far (device-side; the host goes on)."""
        if self._pending is not None:
            self._L.call("pdn_stream_wait_event", self._hp.stream(), self._pending)

    def exposed_wait_events(self):
        """(start, stop) timing events bracketing `wait()` on the compute stream -- what bench.py
        reports as exposed communication."""
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        self._L.call("pdn_event_create", ctypes.byref(e0), 1)
        self._L.call("pdn_event_create", ctypes.byref(e1), 1)
        return e0.value, e1.value

    # -- host-synchronous helpers ---------------------------------------------------------------------
    def barrier(self):
        t = self._hp.zeros((1,), np.float32)
        self.all_reduce(t, SUM)
        self.wait()
        self._hp.synchronize()

    def all_reduce_scalar(self, value: float, op=MAX) -> float:
        t = self._hp.from_numpy(np.array([value], np.float32))
        self.all_reduce(t, op)
        self.wait()
        return float(t.get()[0])

    def destroy(self):
        if self._comm:
            self._hp.synchronize()
            self._L.call("pdn_stream_synchronize", self._stream)
            self._L.call("pdn_comm_destroy", self._comm)
            self._L.call("pdn_stream_destroy", self._stream)
            self._comm = None


_backends = {}      # name -> factory(rank, world): communicators registered from outside the product


def register_backend(name: str, factory):
    """Make `init_process_group(name)` build its communicator with `factory(rank, world)`.  The product ships ONE
    communicator (RCCL through the C ABI); the GPU-less tests register a host one that moves NumPy buffers
    (tests/gloo_comm.py) so that the N > 1 logic of DataParallel runs without a GPU."""
    _backends[name] = factory


def init_process_group(backend=None, device_index=None):
    """Create the process-wide communicator from the launcher's environment (RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT).  backend: "rccl" (alias "nccl"), or a name given to `register_backend`.
    Returns (rank, world)."""
    global _group
    rank, world = rendezvous.env_rank_world()
    if _group is None:
        if backend is None:
            backend = "rccl"
        if backend in ("rccl", "nccl"):
            local = int(os.environ.get("LOCAL_RANK", rank)) if device_index is None else device_index
            _group = RcclComm(rank, world, local)
        elif backend in _backends:
            _group = _backends[backend](rank, world)
        else:
            raise ValueError(f"unknown backend {backend!r} (rccl | nccl" +
                             "".join(f" | {b}" for b in _backends) + ")")
    return _group.rank, _group.world


def get_group():
    return _group


def destroy_process_group():
    global _group
    if _group is not None:
        _group.destroy()
        _group = None


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
        self.module = module
        self.group = process_group if process_group is not None else _group
        self.world = self.group.world if self.group is not None else 1
        self.rank = self.group.rank if self.group is not None else 0
        # `always_reduce` runs the collectives even on a single rank (used to exercise the RCCL path
        # on a one-GPU box); normally a lone rank skips them
        self._comm = self.group is not None and (self.world > 1 or always_reduce)
        self.params = [p for p in module.parameters()][::-1]        # reverse registration order
        if not self.params:
            raise ValueError("DataParallel: module has no trainable parameters")
        self.device = self.params[0].device
        if self._comm and self.device.is_hip != (self.group.backend == "rccl"):
            raise ValueError(f"DataParallel: {self.group.backend} communicator cannot move {self.device} arrays")
        self._flatten(bucket_mb)
        for i, p in enumerate(self.params):
            p._grad_hook = self._make_hook(i)
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
            optimizer._table_key = None               # grads moved: rebuild the chunk table
            if [id(p) for p in optimizer.params][::-1] == [id(p) for p in self.params]:
                optimizer._flat_grad, optimizer._flat_offsets = self.flat, self.offsets[::-1]
                optimizer._flat_views = [p.grad for p in optimizer.params]
        if self._comm:
            self._install_embedding_owner()
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
            # a parameter that fills a bucket by itself (the embedding table: 36.9 MB, and the LAST gradient of
            # backward) reduces alone: what was collected before it -- layer 0's 2.6 MB -- goes out now instead of
            # waiting for it (anything under 1/8 of a bucket rides along: not worth a message of its own)
            if n >= cap and off > start and off - start >= cap // 8:        # (off > start: never an EMPTY bucket, tiny caps)
                self.buckets.append((start, off, first, i))
                start, first = off, i
            end = off + (n + 3) // 4 * 4
            if end - start >= cap or i == len(sizes) - 1:
                self.buckets.append((start, end, first, i + 1))
                start, first = end, i + 1
        self.param_bucket = {}
        for b, (_, _, lo, hi) in enumerate(self.buckets):
            for i in range(lo, hi):
                self.param_bucket[i] = b

    # -- embedding tables: scatter-ASSIGN across ranks --------------------------------------------
    def _install_embedding_owner(self):
        """`fused.embedding` consults `weight._dp_owner(ids)` before scattering: it returns a (V,)
        float32 vector holding, per row, 1 + the highest rank whose shard contains that token id, and
        this rank's own tag; the scatter then skips rows owned by a later rank, so the summed gradient
        equals the single-process scatter-assign on the concatenated batch (tensor.py:937-940)."""
        group, rank, device = self.group, self.rank, self.device

        def owner_of(ids, V):
            xp = device.xp
            with device:
                own = xp.zeros((V,), dtype=np.float32)
                own[ids.reshape(-1)] = float(rank + 1)
            group.all_reduce(own, MAX)
            group.wait()
            return own, float(rank + 1)

        for p in self.params:
            if p.ndim == 2:
                p._dp_owner = owner_of           # used only if the parameter feeds fused.embedding

    # -- hooks ------------------------------------------------------------------------------------
    def _reset(self):
        self._ready = [0] * len(self.buckets)
        self._next = 0

    def _make_hook(self, index):
        def hook(_param):
            b = self.param_bucket[index]
            self._ready[b] += 1
            self._launch_ready()
        return hook

    def _launch_ready(self, force=False):
        if not self._comm:
            return
        while self._next < len(self.buckets):
            lo, hi, plo, phi = self.buckets[self._next]
            if not force and self._ready[self._next] < phi - plo:
                break
            self.group.all_reduce(self.flat[lo:hi], SUM)
            self._next += 1

    def finish(self):
        """Call after backward(): issues any bucket not yet launched (parameters that received no
        gradient this step) and orders the optimizer behind all reductions.  Gradients then hold the
        SUM over ranks; the optimizer divides by the world size through `grad_scale`."""
        if self._comm:
            self._launch_ready(force=True)
            self.group.wait()
        self._reset()

    def zero_grad(self):
        with self.device:
            self.flat[...] = 0.0

    def broadcast_parameters(self, src=0):
        for p in self.module._parameters.values():
            a = p.data
            if not (a.flags.c_contiguous if isinstance(a, np.ndarray) else a.is_contiguous()):
                raise ValueError("DataParallel: non-contiguous parameter")
            self.group.broadcast(a, src)
        self.group.wait()

    def __call__(self, *a, **kw):
        return self.module(*a, **kw)
