"""Dispatch, the device runtime (host memory as device memory, streams / events / graphs as no-ops) and the RCCL stand-in (gloo on host buffers).
(One part of the TEST-ONLY host emulation of the pdnhip C ABI: see tests/abi_emulator/__init__.py.)"""
import ctypes  # noqa: F401
import math  # noqa: F401

import numpy as np  # noqa: F401

from pydynet_amd import _lib  # noqa: F401
from ._base import _NP, _ints, view, flat  # noqa: F401


class RuntimeMixin:
    # address -> owning NumPy block, shared by EVERY emulator instance of the process: arrays of an earlier test that the
    # garbage collector releases late call free() on whichever instance is installed THEN -- with a table per instance
    # their blocks were gone with the old instance, malloc handed the same addresses to the new one, and the late free()
    # dropped a LIVE block (seen as a wrong first loss / a segmentation fault in test_dropin_reference_programs.py after the
    # bench.main() tests).  Shared, an address stays taken until its own owner frees it.
    _blocks = {}

    def __init__(self):
        self.protos = _lib.parse_header()
        self.calls = []
        self._handles = 1000

    # -- dispatch -------------------------------------------------------------------------
    def call(self, name, *args):
        assert name in self.protos, f"{name} is not declared in include/pdn_hip.h"
        assert len(args) == len(self.protos[name][1]), (name, len(args), len(self.protos[name][1]))
        self.calls.append(name)
        rc = getattr(self, name)(*args)
        if rc:
            raise _lib.HipLibraryError(f"{name} failed (emulated, code {rc})", rc)

    def query(self, name, *args):
        assert name in self.protos and len(args) == len(self.protos[name][1]), name
        return getattr(self, name)(*args)

    # -- library ----------------------------------------------------------------------------
    def pdn_device_count(self): return 1
    def pdn_abi_version(self): return 1
    def pdn_stream_synchronize(self, stream): return 0

    # -- device runtime: "HBM" is host memory owned by NumPy arrays kept alive in a table -------------
    def free(self, ptr):                      # the fast path _Buffer.__del__ uses
        self._blocks.pop(int(ptr), None)
        return 0

    def pdn_set_device(self, device): return 0 if device == 0 else 101
    def pdn_device_synchronize(self): return 0

    def pdn_malloc(self, out, nbytes):
        block = np.empty(max(int(nbytes), 1) + 64, np.uint8)
        ptr = (block.ctypes.data + 63) & ~63
        self._blocks[ptr] = block
        ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = ptr
        return 0

    def pdn_free(self, ptr): return self.free(ptr)
    def pdn_empty_cache(self): return 0

    def pdn_mem_stats(self, device, *outs):
        vals = (sum(b.size for b in self._blocks.values()), 0, 0, len(self._blocks), 0, 0)
        for o, v in zip(outs, vals):
            ctypes.cast(o, ctypes.POINTER(ctypes.c_int64))[0] = v
        return 0

    def _copy(self, dst, src, n):
        ctypes.memmove(int(dst), int(src), int(n))
        return 0

    def pdn_memcpy_h2d(self, dst, src, n, stream): return self._copy(dst, src, n)
    def pdn_memcpy_d2h(self, dst, src, n, stream): return self._copy(dst, src, n)
    def pdn_memcpy_d2d(self, dst, src, n, stream): return self._copy(dst, src, n)

    def pdn_host_alloc(self, out, nbytes): return self.pdn_malloc(out, nbytes)
    def pdn_host_free(self, ptr): return self.free(ptr)

    def pdn_host_alloc_mapped(self, host_out, dev_out, nbytes):
        rc = self.pdn_malloc(host_out, nbytes)
        ctypes.cast(dev_out, ctypes.POINTER(ctypes.c_void_p))[0] = ctypes.cast(host_out, ctypes.POINTER(ctypes.c_void_p))[0]
        return rc
    def pdn_memcpy_d2h_async(self, dst, src, n, stream): return self._copy(dst, src, n)

    def pdn_memset(self, dst, value, n, stream):
        ctypes.memset(int(dst), int(value), int(n))
        return 0

    def _handle(self, out):
        self._handles += 1
        ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = self._handles
        return 0

    def pdn_compute_stream(self, out): return self._handle(out)
    def pdn_stream_create(self, out, prio): return self._handle(out)
    def pdn_stream_destroy(self, s): return 0
    def pdn_stream_wait_event(self, s, e): return 0
    def pdn_event_create(self, out, timing): return self._handle(out)
    def pdn_event_record(self, e, s):
        import time
        self.__dict__.setdefault("_etimes", {})[getattr(e, "value", e)] = time.perf_counter()   # the host clock stands in
        return 0
    def pdn_event_synchronize(self, e): return 0
    def pdn_event_destroy(self, e): return 0

    def pdn_event_elapsed_ms(self, a, b, out):
        t = self.__dict__.get("_etimes", {})
        ms = 1e3 * (t.get(getattr(b, "value", b), 0.0) - t.get(getattr(a, "value", a), 0.0))
        ctypes.cast(out, ctypes.POINTER(ctypes.c_float))[0] = max(ms, 1e-6)
        return 0

    # hipGraph capture has no host counterpart: the emulated library refuses it (tests are GPU-only)
    def pdn_pool_create(self, out): return -2
    def pdn_pool_activate(self, pool): return -2 if pool else 0
    def pdn_graph_begin_capture(self, stream): return -2

    # -- RCCL stands in as torch.distributed gloo on the host buffers (collectives run synchronously) --
    def pdn_comm_unique_id(self, out):
        ctypes.memmove(out, bytes(range(128)), 128)
        return 0

    def pdn_comm_init(self, out, rank, world, uid):
        import torch.distributed as dist
        assert bytes(ctypes.string_at(uid, 128) if not isinstance(uid, bytes) else uid[:128]) == bytes(range(128))
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=world)
        return self._handle(out)

    def pdn_comm_destroy(self, comm):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        return 0

    def pdn_comm_allreduce_f32(self, comm, buf, n, op, stream):
        import torch
        import torch.distributed as dist
        dist.all_reduce(torch.from_numpy(flat(buf, n)), op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
        return 0

    def pdn_comm_broadcast(self, comm, buf, nbytes, root, stream):
        import torch
        import torch.distributed as dist
        dist.broadcast(torch.from_numpy(flat(buf, nbytes, np.uint8)), src=root)
        return 0

    def pdn_comm_allgather(self, comm, send, recv, nbytes, stream):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size()
        parts = list(torch.from_numpy(flat(recv, nbytes * world, np.uint8).reshape(world, nbytes)).unbind(0))
        dist.all_gather(parts, torch.from_numpy(np.array(flat(send, nbytes, np.uint8))))
        return 0
    def pdn_gemm_f32_workspace_bytes(self, M, N, K, nb): return 64 * M * N * nb * 4
    def pdn_rmsnorm_bwd_workspace_bytes(self, rows, cols): return 1024 * cols * 4
    def pdn_embedding_scatter_workspace_bytes(self, V): return V * 4
    def pdn_gemm_prof_enable(self, on): return 0
    def pdn_gemm_prof_collect_families(self, ms, fl, n): return 0
    def pdn_gemm_prof_collect_fused(self, ms, fl, by, n): return 0
