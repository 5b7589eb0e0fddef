"""TEST-ONLY host emulation of the pdnhip C ABI.

Lets the whole Python front end (hipnp striding / broadcasting / views / indexing, the tape
engine, nn, optim, data-parallel wrapper) run in the GPU-less container: `hipnp` is pointed
at host memory and every `pdn_*` entry point is answered by a NumPy statement of the same
contract (include/pdn_hip.h).  It is NOT a product path: `pydynet_amd` never imports it, and
on the GPU box the `-m gpu` tests exercise the real library.  Numerics here follow NumPy, so
CPU tests compare against the oracle at fp32 round-off.
"""
import ctypes
import math

import numpy as np

from pydynet_amd import _lib

_NP = {0: np.float32, 1: np.float64, 2: np.int64, 3: np.uint8, 4: np.int32, 5: np.float16}


def _ints(arr, n):
    return [int(arr[i]) for i in range(n)] if n else []


def view(ptr, shape, strides, dtype):
    """NumPy view of host memory at `ptr` with element strides (may be 0 or negative)."""
    dtype = np.dtype(dtype)
    shape, strides = [int(s) for s in shape], [int(s) for s in strides]
    if any(s == 0 for s in shape) or not ptr:
        return np.zeros(shape, dtype)
    lo = sum((s - 1) * st for s, st in zip(shape, strides) if st < 0)
    hi = sum((s - 1) * st for s, st in zip(shape, strides) if st > 0)
    n = hi - lo + 1
    buf = (ctypes.c_char * (n * dtype.itemsize)).from_address(int(ptr) + lo * dtype.itemsize)
    base = np.frombuffer(buf, dtype=dtype)
    return np.lib.stride_tricks.as_strided(base[-lo:], shape=shape, strides=[st * dtype.itemsize for st in strides])


def flat(ptr, n, dtype=np.float32):
    return view(ptr, (n,), (1,), dtype)


class EmulatedLib:
    def __init__(self):
        self.protos = _lib.parse_header()
        self.calls = []
        self._blocks, self._handles = {}, 1000

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

    # -- gemm -------------------------------------------------------------------------------
    def pdn_gemm_f32(self, M, N, K, alpha, A, a_rs, a_cs, B, b_rs, b_cs, beta, C, ldc, bias, nb1, nb2,
                     a1, a2, b1, b2, c1, c2, residual, colsum, colsum_acc, ws, wsb, stream):
        if M == 0 or N == 0 or nb1 == 0 or nb2 == 0:
            return 0
        a = view(A, (nb1, nb2, M, K), (a1, a2, a_rs, a_cs), np.float32)
        b = view(B, (nb1, nb2, K, N), (b1, b2, b_rs, b_cs), np.float32)
        c = view(C, (nb1, nb2, M, N), (c1, c2, ldc, 1), np.float32)
        r = np.float32(alpha) * np.matmul(a, b)
        if bias:
            r = r + flat(bias, N)
        if residual:
            r = r + view(residual, (nb1, nb2, M, N), (c1, c2, ldc, 1), np.float32)
        if colsum:
            assert nb1 * nb2 == 1 and a_rs == 1 and b_cs == 1, "b_colsum: x^T @ g layout only"
            cs = b[0, 0].sum(0)
            flat(colsum, N)[...] = flat(colsum, N) + cs if colsum_acc else cs
        if beta != 0.0:
            r = r + np.float32(beta) * c
        c[...] = r
        return 0

    # -- relu(linear) with one bit per element for the gradient (csrc/gemm.hip: bit c of word [row * (N / 32) + col / 32])
    def pdn_relu_mask_supported(self, rows, cols): return int(rows > 0 and cols > 0 and cols % 32 == 0)

    @staticmethod
    def _bits(ptr, rows, cols):
        return np.unpackbits(view(ptr, (rows, cols // 8), (cols // 8, 1), np.uint8), axis=1, bitorder="little").astype(bool)

    def pdn_linear_relu_fwd_f32(self, x, x_rs, W, w_rs, w_cs, bias, h, ldh, mask, M, N, K, stream):
        assert N % 32 == 0
        z = view(x, (M, K), (x_rs, 1), np.float32) @ view(W, (K, N), (w_rs, w_cs), np.float32)
        if bias:
            z = z + flat(bias, N)
        view(h, (M, N), (ldh, 1), np.float32)[...] = np.maximum(np.float32(0), z)
        view(mask, (M, N // 8), (N // 8, 1), np.uint8)[...] = np.packbits(z >= 0, axis=1, bitorder="little")
        self._count(16)
        return 0

    def pdn_linear_dx_masked_f32(self, g, g_rs, W, w_rs, w_cs, dx, ld, existing, mask, partials, M, fin, fout, stream):
        assert fin % 32 == 0
        r = view(g, (M, fout), (g_rs, 1), np.float32) @ view(W, (fin, fout), (w_rs, w_cs), np.float32).T
        if existing:
            r = r + view(existing, (M, fin), (ld, 1), np.float32)
        r = np.where(self._bits(mask, M, fin), r, np.float32(0)).astype(np.float32)
        view(dx, (M, fin), (ld, 1), np.float32)[...] = r
        if partials:
            nb = (M + 31) // 32
            pad = np.zeros((nb * 32, fin), np.float32)
            pad[:M] = r
            view(partials, (nb, fin), (fin, 1), np.float32)[...] = pad.reshape(nb, 32, fin).sum(1)
        self._count(17)
        return 0

    def pdn_relu_mask_bwd_f32(self, g, mask, dz, rows, cols, stream):
        assert cols % 32 == 0
        gv = np.array(view(g, (rows, cols), (cols, 1), np.float32))
        view(dz, (rows, cols), (cols, 1), np.float32)[...] = np.where(self._bits(mask, rows, cols), gv, np.float32(0))
        return 0

    def pdn_gemm_rowtile_mode(self, mode): return 1       # (kernel selection only: results are bit-identical)

    # -- launch counters per kernel (include/pdn_hip.h: pdn_kernel_counters): the emulated entry points count the kernel
    #    the library would have launched for the same arguments (its dispatch rules restated), so the gates of bench.py
    #    and the path assertions of the tests run without a GPU
    def _count(self, slot):
        self._counters = getattr(self, "_counters", [0] * 24)
        self._counters[slot] += 1

    def pdn_kernel_counters(self, out, n, reset):
        c = getattr(self, "_counters", [0] * 24)
        if out:
            arr = ctypes.cast(out, ctypes.POINTER(ctypes.c_int64))
            for i in range(min(int(n), 24)):
                arr[i] = c[i]
        if reset:
            self._counters = [0] * 24
        return 0

    @staticmethod
    def _rowtile_takes(M, pieces):
        rb = (M + 255) // 256
        return rb >= 192 or rb * pieces >= 256

    @staticmethod
    def _att_p(L, hd):                       # csrc/attention_p.hip: pdn_attention_p_supported (+ attention_blocks.hip:
        # 512 / 768 / 1024 positions as 256-row block pairs on the same kernels; counted once per call here)
        return hd == 48 and L % 32 == 0 and (32 <= L <= 256 or (L % 256 == 0 and L <= 1024))

    def pdn_gemm_rowres_supported(self, M, N, K, lda, ldb, ldc, b_trans):
        return int(K == 288 and N % 32 == 0 and N >= 96 and M >= 1 and lda % 4 == 0 and ldb % 4 == 0 and lda >= K
                   and ldb >= (K if b_trans else N) and ldc >= N and 32 * ldc < (1 << 30))

    def pdn_gemm_rowres_f32(self, A, B, C, bias, residual, M, N, K, lda, ldb, ldc, b_trans, stream):
        if M == 0 or N == 0:
            return 0
        if not self.pdn_gemm_rowres_supported(M, N, K, lda, ldb, ldc, b_trans):
            return -2
        a = view(A, (M, K), (lda, 1), np.float32)
        b = view(B, (K, N), (1, ldb) if b_trans else (ldb, 1), np.float32)
        r = np.matmul(a, b)
        if bias:
            r = r + flat(bias, N)
        if residual:
            r = r + view(residual, (M, N), (ldc, 1), np.float32)
        view(C, (M, N), (ldc, 1), np.float32)[...] = r
        return 0

    def pdn_linear_lse_supported(self, M, V, K):
        return int(K == 288 and V % 32 == 0 and V >= 96 and M >= 49152)

    def pdn_linear_lse_fwd_f32(self, x, w, bias, logits, lse, M, V, K, ldx, ldw, ldl, stream):
        if not self.pdn_linear_lse_supported(M, V, K) or ldl % 4:
            return -2
        z = np.matmul(view(x, (M, K), (ldx, 1), np.float32), view(w, (K, V), (ldw, 1), np.float32))
        if bias:
            z = z + flat(bias, V)
        view(logits, (M, V), (ldl, 1), np.float32)[...] = z
        m = z.max(-1, keepdims=True)
        flat(lse, M)[...] = (m + np.log(np.exp(z - m).sum(-1, keepdims=True)))[:, 0]
        return 0

    def pdn_linear_rowmax_supported(self, M, V, K):
        return int(K == 288 and V % 32 == 0 and V >= 96 and M >= 1)

    def pdn_linear_rowmax_parts(self, M, V, K):
        """As csrc/gemm_rowres.hip: the chunks of 96 columns are split over the grid until the chip is full."""
        if not self.pdn_linear_rowmax_supported(M, V, K):
            return 0
        chunks, row_blocks, nsplit = (V + 95) // 96, (M + 255) // 256, 1
        while row_blocks * nsplit < 256 and nsplit < chunks:
            nsplit += 1
        cpw = (chunks + nsplit - 1) // nsplit
        return (chunks + cpw - 1) // cpw

    def pdn_linear_rowmax_fwd_f32(self, x, w, bias, logits, rowmax, M, V, K, ldx, ldw, ldl, stream):
        self._count(5 if self._rowtile_takes(M, V // 32) else 0)
        if not self.pdn_linear_rowmax_supported(M, V, K) or ldl % 4:
            return -2
        z = np.matmul(view(x, (M, K), (ldx, 1), np.float32), view(w, (K, V), (ldw, 1), np.float32))
        if bias:
            z = z + flat(bias, V)
        view(logits, (M, V), (ldl, 1), np.float32)[...] = z
        parts = self.pdn_linear_rowmax_parts(M, V, K)
        chunks = (V + 95) // 96
        cpw = (chunks + parts - 1) // parts
        out = flat(rowmax, parts * M).reshape(parts, M)
        for q in range(parts):
            out[q] = z[:, q * cpw * 96:min(V, (q + 1) * cpw * 96)].max(-1)
        return 0

    @staticmethod
    def _outres_splits(M, K):
        """pdn_gemm_outres_plan(M, K) (csrc/gemm_outres.hip): ranges of the contraction over the grid (1 = none)."""
        npieces, wg8, wg4 = K // 32, (M + 255) // 256, (M + 127) // 128
        if wg8 >= 224:
            return 1
        s8 = min((256 + wg8 - 1) // wg8, npieces // 24)
        if s8 >= 2 and wg8 * s8 >= 224:
            kps = (npieces + s8 - 1) // s8
            return (npieces + kps - 1) // kps
        if wg4 >= 224:
            return 1
        splits = min((448 + wg4 - 1) // wg4, npieces // 24)
        if splits < 2:
            return 1
        kps = (npieces + splits - 1) // splits
        return (npieces + kps - 1) // kps

    def pdn_linear_ce_dx_deferred_supported(self, rows, V, fin):
        return int(fin == 288 and V % 32 == 0 and V >= 32 and rows >= 1)

    def pdn_linear_ce_dx_deferred_workspace_bytes(self, rows, V, fin):
        s_ = self._outres_splits(rows, V)
        return V * 288 * 4 + (s_ * rows * 289 * 4 if s_ > 1 else 0)       # W^T copy, then the range slabs

    def pdn_linear_ce_dx_deferred_f32(self, logits, rowmax, parts, targets, gscale, W, dx, lse, rows, V, fin, ws, wsb, stream):
        self._count(12)
        if not self.pdn_linear_ce_dx_deferred_supported(rows, V, fin):
            return -2
        if wsb < self.pdn_linear_ce_dx_deferred_workspace_bytes(rows, V, fin):
            return -1
        a = flat(logits, rows * V).reshape(rows, V)
        m = flat(rowmax, parts * rows).reshape(parts, rows).max(0)
        t = np.clip(flat(targets, rows, np.int64), 0, V - 1)
        e = np.exp(a - m[:, None])
        z = e.sum(-1)
        w = flat(W, fin * V).reshape(fin, V)
        flat(dx, rows * fin).reshape(rows, fin)[...] = np.float32(gscale) * ((e @ w.T) / z[:, None] - w.T[t])
        flat(lse, rows)[...] = m + np.log(z)
        return 0

    def pdn_cross_entropy_from_lse_f32(self, logits, ldl, lse, targets, rows, V, mean, loss_row, loss_out, err, stream):
        t = np.array(flat(targets, rows, np.int64))
        bad = (t < 0) | (t >= V)
        if bad.any():
            ctypes.cast(err, ctypes.POINTER(ctypes.c_int))[0] = 1
            t[bad] = 0
        z = view(logits, (rows, V), (ldl, 1), np.float32)
        lr = flat(lse, rows) - z[np.arange(rows), t]
        flat(loss_row, rows)[...] = lr
        flat(loss_out, 1)[0] = lr.sum(dtype=np.float32) * np.float32(1.0 / rows if mean else 1.0)
        return 0

    def pdn_gemm_outres_blocks_supported(self, M, kb, nb):
        K = kb * nb
        big, mid = (M + 255) // 256 >= 224, (M + 127) // 128 >= 224 and K >= 1536
        return int(kb > 0 and kb % 32 == 0 and nb >= 1 and (big or mid))

    def pdn_gemm_outres_blocks_nt_f32(self, A, W, bstride, kb, nb, C, residual, M, lda, ldc, stream):
        if M == 0:
            return 0
        if not self.pdn_gemm_outres_blocks_supported(M, kb, nb) or bstride % 4:
            return -2
        a = view(A, (M, nb * kb), (lda, 1), np.float32)
        r = np.zeros((M, 288), np.float32)
        for i in range(nb):
            w = view(int(W) + 4 * i * bstride, (288, kb), (kb, 1), np.float32)
            r += np.matmul(a[:, i * kb:(i + 1) * kb], w.T)
        if residual:
            r = r + view(residual, (M, 288), (ldc, 1), np.float32)
        view(C, (M, 288), (ldc, 1), np.float32)[...] = r
        return 0

    # -- projections with a fused epilogue (csrc/gemm_rowres.hip, round 4) ----------------------------------
    def pdn_gateup_swiglu_supported(self, M, F, K):
        return int(K == 288 and F % 96 == 0 and F >= 96 and M >= 1 and 64 * F < (1 << 29))

    def pdn_gateup_swiglu_fwd_f32(self, x, wg, w_stride, gu, h, M, F, K, ldx, stream):
        self._count(2 if self._rowtile_takes(M, 2 * (F // 32)) else 6)
        if M == 0 or F == 0:
            return 0
        if not self.pdn_gateup_swiglu_supported(M, F, K) or w_stride % 4 or abs(w_stride) < K * F:
            return -2
        a = view(x, (M, K), (ldx, 1), np.float32)
        g = np.matmul(a, view(wg, (K, F), (F, 1), np.float32))
        u = np.matmul(a, view(int(wg) + 4 * w_stride, (K, F), (F, 1), np.float32))      # (either order in memory)
        out = flat(gu, M * 2 * F).reshape(M, 2 * F)
        out[:, :F], out[:, F:] = g, u
        flat(h, M * F).reshape(M, F)[...] = g / (1 + np.exp(-g)) * u
        return 0

    # -- the same projections with the RMSNorm in front folded into the A load (round 5, csrc/gemm_rowtile.hip NORM) ----
    def _norm_rows(self, x, norm_w, eps, xn, rms, M, K, ldx):
        xv = view(x, (M, K), (ldx, 1), np.float32)
        r = np.sqrt((xv * xv).mean(-1, dtype=np.float32) + np.float32(eps)).astype(np.float32)
        y = (xv / r[:, None] * flat(norm_w, K)).astype(np.float32)
        view(xn, (M, K), (K, 1), np.float32)[...] = y
        flat(rms, M)[...] = r

    def pdn_gateup_swiglu_norm_supported(self, M, F, K):
        return 1 if (self.pdn_gateup_swiglu_supported(M, F, K) and self._rowtile_takes(M, 2 * (F // 32))) else 0

    def pdn_gateup_swiglu_norm_fwd_f32(self, x, norm_w, eps, xn, rms, wg, w_stride, gu, h, M, F, K, ldx, stream):
        if M == 0 or F == 0:
            return 0
        if not self.pdn_gateup_swiglu_norm_supported(M, F, K):
            return -2
        self._norm_rows(x, norm_w, eps, xn, rms, M, K, ldx)
        return self.pdn_gateup_swiglu_fwd_f32(xn, wg, w_stride, gu, h, M, F, K, K, stream)

    def pdn_qkv_rope_norm_supported(self, M, D, K, L, hd):
        return 1 if (self.pdn_qkv_rope_supported(M, D, K, L, hd) and self._rowtile_takes(M, 3 * D // 32)) else 0

    def pdn_qkv_rope_norm_fwd_f32(self, x, norm_w, eps, xn, rms, wq, w_stride, qkv, rope, M, D, K, L, hd, ldx, stream):
        if M == 0 or D == 0:
            return 0
        if not self.pdn_qkv_rope_norm_supported(M, D, K, L, hd):
            return -2
        self._norm_rows(x, norm_w, eps, xn, rms, M, K, ldx)
        return self.pdn_qkv_rope_fwd_f32(xn, wq, w_stride, qkv, rope, M, D, K, L, hd, K, stream)

    def pdn_swiglu_bwd_gemm_f32(self, dy, wd, gu, dgu, M, F, K, ldy, stream):
        self._count(3 if self._rowtile_takes(M, F // 32) else 6)
        if M == 0 or F == 0:
            return 0
        if not self.pdn_gateup_swiglu_supported(M, F, K):
            return -2
        d = np.matmul(view(dy, (M, K), (ldy, 1), np.float32), flat(wd, F * K).reshape(F, K).T)
        a = np.array(flat(gu, M * 2 * F).reshape(M, 2 * F))
        g, u = a[:, :F], a[:, F:]
        sg = 1 / (1 + np.exp(-g))
        out = flat(dgu, M * 2 * F).reshape(M, 2 * F)
        out[:, :F] = d * u * sg * (1 + g * (1 - sg))
        out[:, F:] = d * g * sg
        return 0

    # -- the same two epilogues in the stores of the tiled kernel (other model widths; csrc/gemm.hip SWI) -----------------
    def pdn_gateup_swiglu_tiled_supported(self, M, F, K):
        return int(M >= 4096 and M % 128 == 0 and F >= 256 and F % 32 == 0 and K >= 64 and K % 4 == 0)

    def pdn_gateup_swiglu_tiled_workspace_bytes(self, F, K): return K * ((2 * F + 255) // 256 * 256) * 4

    def pdn_gateup_swiglu_tiled_fwd_f32(self, x, ldx, wg, w_stride, gu, h, M, F, K, ws, wsb, stream):
        if not self.pdn_gateup_swiglu_tiled_supported(M, F, K) or w_stride % 4:
            return -1
        if wsb < self.pdn_gateup_swiglu_tiled_workspace_bytes(F, K):
            return -3
        a = view(x, (M, K), (ldx, 1), np.float32)
        g = np.matmul(a, view(wg, (K, F), (F, 1), np.float32))
        u = np.matmul(a, view(int(wg) + 4 * w_stride, (K, F), (F, 1), np.float32))
        out = flat(gu, M * 2 * F).reshape(M, 2 * F)
        out[:, :F], out[:, F:] = g, u
        flat(h, M * F).reshape(M, F)[...] = g / (1 + np.exp(-g)) * u
        self._count(19)
        return 0

    def pdn_swiglu_bwd_tiled_supported(self, M, F, K):
        return int(M >= 4096 and M % 128 == 0 and F >= 256 and F % 4 == 0 and K >= 64 and K % 4 == 0)

    def pdn_swiglu_bwd_tiled_f32(self, dy, ldy, wd, gu, dgu, M, F, K, stream):
        if not self.pdn_swiglu_bwd_tiled_supported(M, F, K):
            return -1
        d = np.matmul(view(dy, (M, K), (ldy, 1), np.float32), flat(wd, F * K).reshape(F, K).T)
        a = np.array(flat(gu, M * 2 * F).reshape(M, 2 * F))
        g, u = a[:, :F], a[:, F:]
        sg = 1 / (1 + np.exp(-g))
        out = flat(dgu, M * 2 * F).reshape(M, 2 * F)
        out[:, :F] = d * u * sg * (1 + g * (1 - sg))
        out[:, F:] = d * g * sg
        self._count(20)
        return 0

    def pdn_qkv_rope_supported(self, M, D, K, L, hd):
        return int(K == 288 and D % 96 == 0 and hd >= 32 and hd % 2 == 0 and D % hd == 0 and 3 * D < 65536
                   and L % 32 == 0 and L > 0 and M % L == 0)

    def pdn_rope_table_f32(self, cos, sin, out, L, hd, stream):
        c = flat(cos, L * hd // 2).reshape(L, hd // 2)
        s = flat(sin, L * hd // 2).reshape(L, hd // 2)
        t = flat(out, L * hd * 2).reshape(L, hd, 2)
        t[:, :, 0] = np.repeat(c, 2, axis=1)
        t[:, 0::2, 1] = -s
        t[:, 1::2, 1] = s
        return 0

    def pdn_qkv_rope_fwd_f32(self, x, wq, w_stride, qkv, rope, M, D, K, L, hd, ldx, stream):
        self._count(4 if self._rowtile_takes(M, 3 * D // 32) else 6)
        if M == 0 or D == 0:
            return 0
        if not self.pdn_qkv_rope_supported(M, D, K, L, hd) or w_stride % 4:
            return -2
        a = view(x, (M, K), (ldx, 1), np.float32)
        w = [view(int(wq) + 4 * i * w_stride, (K, D), (D, 1), np.float32) for i in range(3)]
        t = flat(rope, L * hd * 2).reshape(L, hd, 2)
        out = flat(qkv, M * 3 * D).reshape(M, 3 * D)
        pos = np.arange(M) % L
        for i in range(3):
            y = np.matmul(a, w[i])
            if i < 2:                               # out = v cos + pair(v) * (-+sin): the table's second entry
                yh = y.reshape(M, D // hd, hd)
                pair = yh.reshape(M, D // hd, hd // 2, 2)[..., ::-1].reshape(M, D // hd, hd)
                y = (yh * t[pos, None, :, 0] + pair * t[pos, None, :, 1]).reshape(M, D)
            out[:, i * D:(i + 1) * D] = y
        return 0

    def pdn_gemm_outres_supported(self, M, N, K, lda, ldb, ldc, b_trans):
        return int(N == 288 and K >= 32 and K % 32 == 0 and M >= 1 and lda % 4 == 0 and ldb % 4 == 0 and lda >= K
                   and ldb >= (K if b_trans else N) and ldc >= N and 288 * ldb < (1 << 30) and 32 * ldc < (1 << 30))

    def pdn_gemm_outres_plan(self, M, K, nw, kps): return 1
    def pdn_gemm_outres_workspace_bytes(self, M, K): return 0

    def pdn_gemm_outres_ws_f32(self, A, B, C, bias, residual, M, N, K, lda, ldb, ldc, b_trans, ws, wsb, stream):
        return self.pdn_gemm_outres_f32(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, b_trans, stream)

    def pdn_gemm_outres_f32(self, A, B, C, bias, residual, M, N, K, lda, ldb, ldc, b_trans, stream):
        if M == 0 or N == 0:
            return 0
        if not self.pdn_gemm_outres_supported(M, N, K, lda, ldb, ldc, b_trans):
            return -2
        a = view(A, (M, K), (lda, 1), np.float32)
        b = view(B, (K, N), (1, ldb) if b_trans else (ldb, 1), np.float32)
        r = np.matmul(a, b)
        if bias:
            r = r + flat(bias, N)
        if residual:
            r = r + view(residual, (M, N), (ldc, 1), np.float32)
        view(C, (M, N), (ldc, 1), np.float32)[...] = r
        return 0

    def pdn_gemm_f64(self, M, N, K, alpha, A, a_rs, a_cs, B, b_rs, b_cs, beta, C, ldc, nb1, nb2,
                     a1, a2, b1, b2, c1, c2, stream):
        if M == 0 or N == 0 or nb1 == 0 or nb2 == 0:
            return 0
        a = view(A, (nb1, nb2, M, K), (a1, a2, a_rs, a_cs), np.float64)
        b = view(B, (nb1, nb2, K, N), (b1, b2, b_rs, b_cs), np.float64)
        c = view(C, (nb1, nb2, M, N), (c1, c2, ldc, 1), np.float64)
        r = alpha * np.matmul(a, b)
        c[...] = r + beta * c if beta != 0.0 else r
        return 0

    # -- elementwise --------------------------------------------------------------------------
    def pdn_ew_binary(self, dt, op, mode, ndim, shape, a, sa, b, sb, scalar, out, so, stream):
        shp = _ints(shape, ndim)
        T = _NP[dt]
        x = view(a, shp, _ints(sa, ndim), T)
        y = view(b, shp, _ints(sb, ndim), T) if mode == 0 else np.asarray(scalar).astype(T)
        if mode == 2:
            x, y = y, x
        if T == np.uint8:
            x, y = x.astype(bool), (y.astype(bool) if isinstance(y, np.ndarray) else y)
        f = {0: np.add, 1: np.subtract, 2: np.multiply, 3: np.divide, 4: np.power, 5: np.maximum,
             6: np.minimum, 16: np.equal, 17: np.not_equal, 18: np.less, 19: np.less_equal,
             20: np.greater, 21: np.greater_equal}[op]
        with np.errstate(all="ignore"):
            r = f(x, y)
        o = view(out, shp, _ints(so, ndim), np.uint8 if op >= 16 else T)
        o[...] = r
        return 0

    def pdn_ew_unary(self, dt, op, ndim, shape, a, sa, out, so, stream):
        shp = _ints(shape, ndim)
        T = _NP[dt]
        x = view(a, shp, _ints(sa, ndim), T)
        one = T(1)

        def sig(v):
            r = np.zeros(v.shape, v.dtype); m = v > 0
            r[m] = 1 / (1 + np.exp(-v[m])); r[~m] = 1 - 1 / (1 + np.exp(v[~m])); return r

        def tnh(v):
            r = np.zeros(v.shape, v.dtype); m = v > 0
            r[m] = 2 / (1 + np.exp(-2 * v[m])) - 1; r[~m] = 1 - 2 / (1 + np.exp(2 * v[~m])); return r

        f = {0: lambda v: v, 1: np.negative, 2: np.exp, 3: np.log, 4: np.abs, 5: np.sign, 6: np.sqrt,
             7: np.square, 8: lambda v: one / v, 9: sig, 10: tnh}[op]
        with np.errstate(all="ignore"):
            r = f(np.array(x))
        view(out, shp, _ints(so, ndim), T)[...] = r
        return 0

    def pdn_cast(self, sdt, ddt, ndim, shape, a, sa, out, so, stream):
        shp = _ints(shape, ndim)
        src = np.array(view(a, shp, _ints(sa, ndim), _NP[sdt]))
        dst = view(out, shp, _ints(so, ndim), _NP[ddt])
        if ddt == 3:
            dst[...] = (src != 0).astype(np.uint8)
        else:
            with np.errstate(all="ignore"):
                dst[...] = src.astype(_NP[ddt])
        return 0

    def pdn_fill(self, dt, value, ndim, shape, out, so, stream):
        view(out, _ints(shape, ndim), _ints(so, ndim), _NP[dt])[...] = np.asarray(value).astype(_NP[dt])
        return 0

    def pdn_masked_fill(self, dt, value, ndim, shape, mask, sm, out, so, stream):
        shp = _ints(shape, ndim)
        m = view(mask, shp, _ints(sm, ndim), np.uint8)
        o = view(out, shp, _ints(so, ndim), _NP[dt])
        o[m != 0] = np.asarray(value).astype(_NP[dt])
        return 0

    def pdn_reduce(self, dt, op, ndim, shape, strides, flags, x, out, ws, wsb, stream):
        shp = _ints(shape, ndim)
        a = np.array(view(x, shp, _ints(strides, ndim), _NP[dt]))
        axes = tuple(i for i in range(ndim) if flags[i])
        kept = [s for i, s in enumerate(shp) if i not in axes]
        if op in (4, 5):
            moved = np.moveaxis(a, axes, list(range(ndim - len(axes), ndim))).reshape(kept + [-1]) if axes else a.reshape(kept + [1])
            r = (np.argmax if op == 4 else np.argmin)(moved, axis=-1)
            flat(out, max(int(np.prod(kept)), 1), np.int64)[...] = np.asarray(r).reshape(-1)
            return 0
        f = {0: np.sum, 1: np.mean, 2: np.max, 3: np.min}[op]
        r = f(a, axis=axes) if axes else a
        flat(out, max(int(np.prod(kept)), 1), _NP[dt])[...] = np.asarray(r, dtype=_NP[dt]).reshape(-1)
        return 0

    # -- fused ----------------------------------------------------------------------------------
    def pdn_softmax_fwd_f32(self, x, y, rows, cols, divisor, causal_L, start_pos, stream):
        a = np.array(flat(x, rows * cols).reshape(rows, cols)) / np.float32(divisor)
        if causal_L > 0:
            r = (np.arange(rows) % causal_L)[:, None] + start_pos
            a = np.where(np.arange(cols)[None, :] > r, -np.inf, a).astype(np.float32)
        e = np.exp(a - a.max(-1, keepdims=True))
        flat(y, rows * cols).reshape(rows, cols)[...] = e / e.sum(-1, keepdims=True)
        return 0

    def pdn_softmax_bwd_f32(self, y, dy, dx, rows, cols, divisor, stream):
        p = np.array(flat(y, rows * cols).reshape(rows, cols))
        g = np.array(flat(dy, rows * cols).reshape(rows, cols))
        flat(dx, rows * cols).reshape(rows, cols)[...] = (g - (g * p).sum(-1, keepdims=True)) * p / np.float32(divisor)
        return 0

    def pdn_rmsnorm_fwd_f32(self, x, w, y, rms, rows, cols, eps, stream):
        a = flat(x, rows * cols).reshape(rows, cols)
        r = np.sqrt((a * a).mean(-1, keepdims=True) + np.float32(eps))
        flat(y, rows * cols).reshape(rows, cols)[...] = a / r * flat(w, cols)
        if rms:
            flat(rms, rows)[...] = r[:, 0]
        return 0

    def pdn_rmsnorm_bwd_f32(self, x, w, rms, dy, res, dx, dw, acc, rows, cols, ws, wsb, stream):
        a = flat(x, rows * cols).reshape(rows, cols)
        g = np.array(flat(dy, rows * cols).reshape(rows, cols))
        r = flat(rms, rows)[:, None]
        z = a / r
        dz = g * flat(w, cols)
        d = (dz - z * (z * dz).mean(-1, keepdims=True)) / r
        if res:
            d = d + flat(res, rows * cols).reshape(rows, cols)
        flat(dx, rows * cols).reshape(rows, cols)[...] = d
        if dw:
            s = (g * z).sum(0)
            flat(dw, cols)[...] = flat(dw, cols) + s if acc else s
        return 0

    def pdn_swiglu_fwd_f32(self, g, u, y, n, stream):
        a = flat(g, n)
        r = a / (1 + np.exp(-a))
        flat(y, n)[...] = r * flat(u, n) if u else r
        return 0

    def pdn_swiglu_bwd_f32(self, g, u, dy, dg, du, n, stream):
        a, d = np.array(flat(g, n)), np.array(flat(dy, n))
        s = 1 / (1 + np.exp(-a))
        r = d * s * (1 + a * (1 - s))
        if u:
            uu = np.array(flat(u, n))
            flat(du, n)[...] = d * a * s
            r = r * uu
        flat(dg, n)[...] = r
        return 0

    def pdn_swiglu_rows_fwd_f32(self, gu, y, rows, F, stream):
        a = flat(gu, rows * 2 * F).reshape(rows, 2 * F)
        g, u = a[:, :F], a[:, F:]
        flat(y, rows * F).reshape(rows, F)[...] = g / (1 + np.exp(-g)) * u
        return 0

    def pdn_swiglu_rows_bwd_f32(self, gu, dy, dgu, rows, F, stream):
        a = np.array(flat(gu, rows * 2 * F).reshape(rows, 2 * F))
        g, u, d = a[:, :F], a[:, F:], flat(dy, rows * F).reshape(rows, F)
        sg = 1 / (1 + np.exp(-g))
        out = flat(dgu, rows * 2 * F).reshape(rows, 2 * F)
        out[:, :F] = d * u * sg * (1 + g * (1 - sg))
        out[:, F:] = d * g * sg
        return 0

    def pdn_relu_bwd_f32(self, x, dy, dx, n, stream):
        a = flat(x, n)
        flat(dx, n)[...] = np.where(np.maximum(0, a) == a, flat(dy, n), 0)
        return 0

    def pdn_rope_f32(self, x, cos, sin, y, rows, L, heads, hd, backward, stream):
        half = hd // 2
        a = np.array(flat(x, rows * heads * hd).reshape(rows, heads, half, 2))
        pos = np.arange(rows) % L
        c = flat(cos, L * half).reshape(L, half)[pos][:, None, :]
        s = flat(sin, L * half).reshape(L, half)[pos][:, None, :] * (-1 if backward else 1)
        o = flat(y, rows * heads * hd).reshape(rows, heads, half, 2)
        r, i = a[..., 0], a[..., 1]
        o[..., 0] = r * c - i * s
        o[..., 1] = r * s + i * c
        return 0

    def pdn_embedding_gather_f32(self, W, V, D, rs, ids, n, out, err, stream):
        w = view(W, (V, D), (rs, 1), np.float32)
        flat(out, n * D).reshape(n, D)[...] = w[flat(ids, n, np.int64)]
        return 0

    def pdn_embedding_scatter_f32(self, g, ids, n, dW, V, D, mode, owner, tag, ws, wsb, stream):
        w = flat(dW, V * D).reshape(V, D)
        gg = np.array(flat(g, n * D).reshape(n, D))
        idx = np.array(flat(ids, n, np.int64))
        if owner and mode != 2:
            # last occurrence first (over ALL local rows), then the data-parallel owner filter
            last = {int(i): r for r, i in enumerate(idx)}
            rows = [r for i, r in sorted(last.items()) if flat(owner, V)[i] == np.float32(tag)]
            idx, gg = idx[rows], gg[rows]
        if mode == 0:
            w[idx] = gg
        elif mode == 1:
            tmp = np.zeros((V, D), np.float32); tmp[idx] = gg
            w += tmp
        else:
            np.add.at(w, idx, gg)
        return 0

    def pdn_take_cols_f32(self, x, n, C, rs, idx, out, err, stream):
        a = view(x, (n, C), (rs, 1), np.float32)
        flat(out, n)[...] = a[np.arange(n), flat(idx, n, np.int64)]
        return 0

    def pdn_put_cols_f32(self, g, idx, dx, n, C, stream):
        flat(dx, n * C).reshape(n, C)[np.arange(n), flat(idx, n, np.int64)] = flat(g, n)
        return 0

    def pdn_cross_entropy_fwd_f32(self, logits, targets, rows, V, mean, loss_row, lse_row, loss_out, err, stream):
        a = flat(logits, rows * V).reshape(rows, V)
        t = flat(targets, rows, np.int64)
        m = a.max(-1, keepdims=True)
        lse = (np.log(np.exp(a - m).sum(-1, keepdims=True)) + m)[:, 0]
        lr = lse - a[np.arange(rows), t]
        flat(lse_row, rows)[...] = lse
        flat(loss_row, rows)[...] = lr
        flat(loss_out, 1)[0] = lr.mean() if mean else lr.sum()
        return 0

    def pdn_cross_entropy_bwd_f32(self, logits, targets, lse_row, upstream, gscale, dlogits, rows, V, stream):
        a = np.array(flat(logits, rows * V).reshape(rows, V))
        t = flat(targets, rows, np.int64)
        sm = np.exp(a - flat(lse_row, rows)[:, None])
        sm[np.arange(rows), t] -= 1
        gs = np.float32(gscale) * (flat(upstream, 1)[0] if upstream else np.float32(1))
        flat(dlogits, rows * V).reshape(rows, V)[...] = sm * gs
        return 0

    def pdn_linear_ce_supported(self, rows, V, fin):
        return int(fin == 288 and V % 32 == 0 and V >= 32 and rows % 32 == 0 and rows >= 32 and 288 * V < (1 << 30))

    def pdn_linear_ce_workspace_bytes(self, rows, V, fin):
        return 2 * (fin + 1) * V * 4 if self.pdn_linear_ce_supported(rows, V, fin) else 0

    def pdn_linear_ce_backward_f32(self, x, ldx, logits, lse, targets, gscale, upstream, W, dx, dx_res, dW, dw_beta,
                                   dbias, db_beta, rows, V, fin, ws, wsb, stream):
        self._count(13)
        if not self.pdn_linear_ce_supported(rows, V, fin):
            return -2
        a = np.array(flat(logits, rows * V).reshape(rows, V))
        t = flat(targets, rows, np.int64)
        sm = np.exp(a - flat(lse, rows)[:, None])
        sm[np.arange(rows), t] -= 1
        d = sm * (np.float32(gscale) * (flat(upstream, 1)[0] if upstream else np.float32(1)))
        xv = view(x, (rows, fin), (ldx, 1), np.float32)
        w = flat(W, fin * V).reshape(fin, V)
        if dx:
            r = d @ w.T
            if dx_res:
                r = r + flat(dx_res, rows * fin).reshape(rows, fin)
            flat(dx, rows * fin).reshape(rows, fin)[...] = r
        if dW:
            g = flat(dW, fin * V).reshape(fin, V)
            g[...] = np.float32(dw_beta) * g + xv.T @ d if dw_beta != 0.0 else xv.T @ d
        if dbias:
            bg = flat(dbias, V)
            bg[...] = np.float32(db_beta) * bg + d.sum(0) if db_beta != 0.0 else d.sum(0)
        return 0

    # -- fused attention (NumPy statement of the same contract) -------------------------------------
    @staticmethod
    def _att_views(ptrs, B, H, L, hd, rs, bs):
        return [view(p, (B, H, L, hd), (bs, hd, rs, 1), np.float32) for p in ptrs]

    def _att_probs(self, q, k, L, hd, causal):
        s = np.matmul(q, k.swapaxes(-1, -2)) / np.float32(math.sqrt(hd))
        if causal:
            s = s + np.triu(np.full((L, L), -np.inf, np.float32), 1)
        m = s.max(-1, keepdims=True)
        e = np.exp(s - m)
        l = e.sum(-1, keepdims=True)
        return e / l, (m + np.log(l))[..., 0]

    def pdn_attention_supported(self, L, hd):
        if hd == 128 and 1 <= L <= 1024:          # csrc/attention_hd128.hip: any length (rows beyond L masked inside)
            return 1
        return 1 if (hd in (48, 64) and L % 32 == 0 and 32 <= L <= 1024) else 0
    def pdn_attention_lds_bytes(self, L, hd): return min(L, 256) * (hd + 4 + 64) * 4
    def pdn_attention_bwd_lds_bytes(self, L, hd): return min(L, 256) * (2 * 68 + 2) * 4

    @staticmethod
    def _rot(a, cos, sin, L, hd, sign):
        """RoPE on (B, H, L, hd) host arrays; cos/sin pointers to (L, hd/2) tables or None."""
        if not cos:
            return a
        c = flat(cos, L * hd // 2).reshape(1, 1, L, hd // 2)
        s = sign * flat(sin, L * hd // 2).reshape(1, 1, L, hd // 2)
        out = np.empty_like(a)
        out[..., 0::2] = a[..., 0::2] * c - a[..., 1::2] * s
        out[..., 1::2] = a[..., 0::2] * s + a[..., 1::2] * c
        return out

    def pdn_attention_persistent_supported(self, L, hd): return int(bool(self.pdn_attention_supported(L, hd)) and self._att_p(L, hd))

    def pdn_rope_rows_f32(self, x, cos, sin, y, rows, L, heads, hd, x_rs, y_rs, backward, stream):
        xv = np.array(view(x, (rows, heads, hd), (x_rs, hd, 1), np.float32))
        c = flat(cos, L * hd // 2).reshape(L, hd // 2)[np.arange(rows) % L][:, None, :]
        s = flat(sin, L * hd // 2).reshape(L, hd // 2)[np.arange(rows) % L][:, None, :] * (-1.0 if backward else 1.0)
        out = np.empty_like(xv)
        out[..., 0::2] = xv[..., 0::2] * c - xv[..., 1::2] * s
        out[..., 1::2] = xv[..., 0::2] * s + xv[..., 1::2] * c
        view(y, (rows, heads, hd), (y_rs, hd, 1), np.float32)[...] = out
        return 0

    def pdn_attention_fwd_f32(self, q, k, v, o, lse, B, H, L, hd, rs, bs, ors, obs, causal, rc, rsn, stream):
        self._count(7 if (not rc and self._att_p(L, hd)) else 9)
        if not self.pdn_attention_supported(L, hd):
            return -2
        Q, K, V = self._att_views([q, k, v], B, H, L, hd, rs, bs)
        O, = self._att_views([o], B, H, L, hd, ors, obs)
        Q, K = self._rot(np.array(Q), rc, rsn, L, hd, 1.0), self._rot(np.array(K), rc, rsn, L, hd, 1.0)
        p, ls = self._att_probs(np.array(Q), np.array(K), L, hd, causal)
        O[...] = np.matmul(p, np.array(V))
        flat(lse, B * H * L).reshape(B, H, L)[...] = ls
        return 0

    def pdn_attention_bwd_workspace_bytes(self, B, H, L): return B * H * L * 4

    def pdn_attention_bwd_f32(self, q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, rs, bs, ors, obs, causal, rc, rsn,
                              ws, wsb, stream):
        self._count(8 if (not rc and self._att_p(L, hd)) else 10)
        if not self.pdn_attention_supported(L, hd):
            return -2
        Q, K, V = [np.array(a) for a in self._att_views([q, k, v], B, H, L, hd, rs, bs)]
        O, DO = [np.array(a) for a in self._att_views([o, do], B, H, L, hd, ors, obs)]
        DQ, DK, DV = self._att_views([dq, dk, dv], B, H, L, hd, rs, bs)
        Q, K = self._rot(Q, rc, rsn, L, hd, 1.0), self._rot(K, rc, rsn, L, hd, 1.0)
        s = np.matmul(Q, K.swapaxes(-1, -2)) / np.float32(math.sqrt(hd))
        p = np.exp(s - flat(lse, B * H * L).reshape(B, H, L, 1))
        if causal:
            p = p * np.tril(np.ones((L, L), np.float32))
        delta = (DO * O).sum(-1, keepdims=True)
        dp = np.matmul(DO, V.swapaxes(-1, -2))
        ds = p * (dp - delta) / np.float32(math.sqrt(hd))
        DV[...] = np.matmul(p.swapaxes(-1, -2), DO)
        DQ[...] = self._rot(np.matmul(ds, K), rc, rsn, L, hd, -1.0)
        DK[...] = self._rot(np.matmul(ds.swapaxes(-1, -2), Q), rc, rsn, L, hd, -1.0)
        return 0

    def _key_bias(self, kb, kbs, B, L):
        return view(kb, (B, 1, 1, L), (kbs, 0, 0, 1), np.float32)

    def pdn_attention_fwd_bias_f32(self, q, k, v, o, lse, B, H, L, hd, rs, bs, ors, obs, causal, kb, kbs, stream):
        self._count(9)
        if not self.pdn_attention_supported(L, hd):
            return -2
        Q, K, V = [np.array(a) for a in self._att_views([q, k, v], B, H, L, hd, rs, bs)]
        O, = self._att_views([o], B, H, L, hd, ors, obs)
        s = np.matmul(Q, K.swapaxes(-1, -2)) / np.float32(math.sqrt(hd)) + self._key_bias(kb, kbs, B, L)
        if causal:
            s = s + np.triu(np.full((L, L), -np.inf, np.float32), 1)
        m = s.max(-1, keepdims=True)
        e = np.exp(s - m)
        z = e.sum(-1, keepdims=True)
        O[...] = np.matmul(e / z, V)
        flat(lse, B * H * L).reshape(B, H, L)[...] = (m + np.log(z))[..., 0]
        return 0

    def pdn_attention_bwd_bias_f32(self, q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, rs, bs, ors, obs, causal, kb, kbs,
                                   ws, wsb, stream):
        self._count(10)
        if not self.pdn_attention_supported(L, hd):
            return -2
        Q, K, V = [np.array(a) for a in self._att_views([q, k, v], B, H, L, hd, rs, bs)]
        O, DO = [np.array(a) for a in self._att_views([o, do], B, H, L, hd, ors, obs)]
        DQ, DK, DV = self._att_views([dq, dk, dv], B, H, L, hd, rs, bs)
        s = np.matmul(Q, K.swapaxes(-1, -2)) / np.float32(math.sqrt(hd)) + self._key_bias(kb, kbs, B, L)
        p = np.exp(s - flat(lse, B * H * L).reshape(B, H, L, 1))
        if causal:
            p = p * np.tril(np.ones((L, L), np.float32))
        delta = (DO * O).sum(-1, keepdims=True)
        dp = np.matmul(DO, V.swapaxes(-1, -2))
        ds = p * (dp - delta) / np.float32(math.sqrt(hd))
        DV[...] = np.matmul(p.swapaxes(-1, -2), DO)
        DQ[...] = np.matmul(ds, K)
        DK[...] = np.matmul(ds.swapaxes(-1, -2), Q)
        return 0

    def pdn_attention_bwd_rotated_f32(self, q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, rs, bs, ors, obs, causal, rc, rsn,
                                      ws, wsb, stream):
        """q, k already rotated: nothing rotated on the way in, dq / dk rotated back on the way out.
        (counted by the plain backward it is stated with: rotation-free operands)"""
        if not self.pdn_attention_supported(L, hd):
            return -2
        rc2 = self.pdn_attention_bwd_f32(q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, rs, bs, ors, obs, causal, None, None,
                                         ws, wsb, stream)
        if rc2:
            return rc2
        DQ, DK = self._att_views([dq, dk], B, H, L, hd, rs, bs)
        DQ[...] = self._rot(np.array(DQ), rc, rsn, L, hd, -1.0)
        DK[...] = self._rot(np.array(DK), rc, rsn, L, hd, -1.0)
        return 0

    # -- general streaming attention -----------------------------------------------------------------
    def pdn_attention_stream_supported(self, hd): return 1 if hd in (16, 24, 32, 48, 64, 96, 128) else 0
    def pdn_attention_stream_bwd_workspace_bytes(self, B, H, Lq): return 4 * B * H * Lq

    def _stream_scores(self, Q, K, B, H, Lq, Lk, hd, causal, start, mask, sb, sh, sq, sk):
        s = np.matmul(Q, K.swapaxes(-1, -2)) / np.float32(math.sqrt(hd))
        if causal:
            qi, ki = np.arange(Lq)[:, None], np.arange(Lk)[None, :]
            s = s + np.where(ki > qi + start, -np.inf, 0.0).astype(np.float32)
        if mask:
            s = s + view(mask, (B, H, Lq, Lk), (sb, sh, sq, sk), np.float32)
        return s

    def pdn_attention_stream_fwd_f32(self, q, k, v, o, lse, B, H, Lq, Lk, hd, qrs, qbs, krs, kbs, causal, start,
                                     mask, sb, sh, sq, sk, rc, rsn, stream):
        self._count(11)
        if not self.pdn_attention_stream_supported(hd) or (rc and start):
            return -2
        Q, O = [view(p, (B, H, Lq, hd), (qbs, hd, qrs, 1), np.float32) for p in (q, o)]
        K, V = [np.array(view(p, (B, H, Lk, hd), (kbs, hd, krs, 1), np.float32)) for p in (k, v)]
        Q, K = self._rot(np.array(Q), rc, rsn, Lq, hd, 1.0), self._rot(K, rc, rsn, Lk, hd, 1.0)
        s = self._stream_scores(Q, K, B, H, Lq, Lk, hd, causal, start, mask, sb, sh, sq, sk)
        m = s.max(-1, keepdims=True)
        e = np.exp(s - m)
        l = e.sum(-1, keepdims=True)
        O[...] = np.matmul(e / l, V)
        flat(lse, B * H * Lq).reshape(B, H, Lq)[...] = (m + np.log(l))[..., 0]
        return 0

    def pdn_attention_stream_bwd_f32(self, q, k, v, o, do, lse, dq, dk, dv, B, H, Lq, Lk, hd, qrs, qbs, krs, kbs,
                                     causal, start, mask, sb, sh, sq, sk, rc, rsn, ws, wsb, stream):
        self._count(11)
        if not self.pdn_attention_stream_supported(hd) or (rc and start):
            return -2
        qv = lambda p: view(p, (B, H, Lq, hd), (qbs, hd, qrs, 1), np.float32)
        kv = lambda p: view(p, (B, H, Lk, hd), (kbs, hd, krs, 1), np.float32)
        Q, O, DO, K, V = np.array(qv(q)), np.array(qv(o)), np.array(qv(do)), np.array(kv(k)), np.array(kv(v))
        Q, K = self._rot(Q, rc, rsn, Lq, hd, 1.0), self._rot(K, rc, rsn, Lk, hd, 1.0)
        s = self._stream_scores(Q, K, B, H, Lq, Lk, hd, causal, start, mask, sb, sh, sq, sk)
        p = np.exp(s - flat(lse, B * H * Lq).reshape(B, H, Lq, 1))
        delta = (DO * O).sum(-1, keepdims=True)
        dp = np.matmul(DO, V.swapaxes(-1, -2))
        ds = p * (dp - delta) / np.float32(math.sqrt(hd))
        kv(dv)[...] = np.matmul(p.swapaxes(-1, -2), DO)
        qv(dq)[...] = self._rot(np.matmul(ds, K), rc, rsn, Lq, hd, -1.0)
        kv(dk)[...] = self._rot(np.matmul(ds.swapaxes(-1, -2), Q), rc, rsn, Lk, hd, -1.0)
        return 0

    def pdn_attention_decode_f32(self, q, kc, vc, o, B, H, T, hd, cbs, stream):
        D = H * hd
        Q = flat(q, B * D).reshape(B, H, hd)
        K = view(kc, (B, T, H, hd), (cbs, D, hd, 1), np.float32)
        V = view(vc, (B, T, H, hd), (cbs, D, hd, 1), np.float32)
        s = np.einsum("bhd,bthd->bht", Q, K) / np.float32(math.sqrt(hd))
        e = np.exp(s - s.max(-1, keepdims=True))
        pr = e / e.sum(-1, keepdims=True)
        flat(o, B * D).reshape(B, H, hd)[...] = np.einsum("bht,bthd->bhd", pr, V)
        return 0

    # -- graph-replayable decode step (csrc/decode.hip) -----------------------------------------------
    def pdn_decode_gemv_f32(self, x, x_rs, norm_w, eps, W, w_rs, blk_cols, w_bs, bias, residual, r_rs, y, y_rs,
                            B, K, N, act, act_ns, act_hd, blk_max, blk_arg, stream):
        if B > 8 or B * K > 16384 or N % blk_cols or blk_cols % 4:
            return -1
        if act == 2:
            H = K // act_hd
            R = np.array(view(x, (B, act_ns, H, 4 + act_hd), (x_rs, H * (4 + act_hd), 4 + act_hd, 1), np.float32))
            m, l, o = R[..., 0], R[..., 1], R[..., 4:]
            m = np.where(l > 0, m, -np.inf)
            w = np.where(l > 0, np.exp(m - m.max(1, keepdims=True)), 0).astype(np.float32)
            a = ((w[..., None] * o).sum(1) / (w * l).sum(1)[..., None]).reshape(B, K)
            X = None
        else:
            X = np.array(view(x, (B, 2 * K if act else K), (x_rs, 1), np.float32))
        if act == 2:
            pass
        elif act:
            g, u = X[:, :K], X[:, K:]
            a = g / (np.float32(1) + np.exp(-g)) * u
        elif norm_w:
            a = X / np.sqrt((X * X).mean(-1, keepdims=True) + np.float32(eps)) * flat(norm_w, K)
        else:
            a = X
        nb = N // blk_cols
        Wv = view(W, (nb, K, blk_cols), (w_bs, w_rs, 1), np.float32)
        out = np.concatenate([a @ Wv[j] for j in range(nb)], axis=1).astype(np.float32)
        if bias:
            out = out + flat(bias, N)
        if residual:
            out = out + view(residual, (B, N), (r_rs, 1), np.float32)
        view(y, (B, N), (y_rs, 1), np.float32)[...] = out
        if blk_max:
            nb_ = self.pdn_decode_gemv_blocks(N)
            tn = -(-N // nb_)
            tn = 16 if N <= 4096 else (32 if N <= 16384 else 64)
            bm, ba = flat(blk_max, B * nb_).reshape(B, nb_), flat(blk_arg, B * nb_, np.int32).reshape(B, nb_)
            for j in range(nb_):
                seg = out[:, j * tn:(j + 1) * tn]
                bm[:, j] = seg.max(-1)
                ba[:, j] = j * tn + seg.argmax(-1)
        return 0

    def pdn_decode_gemv_blocks(self, N):
        return (N + 15) // 16 if N <= 4096 else ((N + 31) // 32 if N <= 16384 else (N + 63) // 64)

    def pdn_decode_attention_f32(self, qkv, rs, cos, sin, kc, vc, parts, B, H, hd, NS, cbs, pos, max_len, stream):
        D, half = H * hd, hd // 2
        p = int(flat(pos, 1, np.int32)[0])
        assert 0 <= p < max_len
        rows = view(qkv, (B, 3 * D), (rs, 1), np.float32)
        c, s_ = flat(cos + 4 * p * half, half), flat(sin + 4 * p * half, half)

        def rot(v):
            a = np.array(v).reshape(H, half, 2)
            out = np.empty_like(a)
            out[..., 0] = a[..., 0] * c - a[..., 1] * s_
            out[..., 1] = a[..., 0] * s_ + a[..., 1] * c
            return out.reshape(D)
        Q = np.stack([rot(rows[b, :D]) for b in range(B)]).reshape(B, H, hd)
        for b in range(B):
            view(kc + 4 * (b * cbs + p * D), (D,), (1,), np.float32)[...] = rot(rows[b, D:2 * D])
            view(vc + 4 * (b * cbs + p * D), (D,), (1,), np.float32)[...] = rows[b, 2 * D:]
        T = p + 1
        K = view(kc, (B, T, H, hd), (cbs, D, hd, 1), np.float32)
        V = view(vc, (B, T, H, hd), (cbs, D, hd, 1), np.float32)
        s = np.einsum("bhd,bthd->bht", Q, K) / np.float32(math.sqrt(hd))
        out = flat(parts, B * NS * H * (4 + hd)).reshape(B, NS, H, 4 + hd)
        chunk = -(-T // NS)
        for sp in range(NS):
            t0, t1 = sp * chunk, min(T, (sp + 1) * chunk)
            if t0 >= t1:
                out[:, sp, :, 0], out[:, sp, :, 1] = -np.inf, 0.0
                continue
            ss = s[:, :, t0:t1]
            m = ss.max(-1)
            e = np.exp(ss - m[..., None])
            out[:, sp, :, 0], out[:, sp, :, 1] = m, e.sum(-1)
            out[:, sp, :, 4:] = np.einsum("bht,bthd->bhd", e, V[:, t0:t1])
        return 0

    def pdn_decode_attention_oproj_f32(self, qkv, rs, cos, sin, kc, vc, Wo, wo_rs, recs, B, H, hd, NS, cbs, pos,
                                       max_len, stream):
        D = H * hd
        if D > 1024 or NS * H > 256:
            return -1
        tmp = np.zeros(B * NS * H * (4 + hd), np.float32)
        rc = self.pdn_decode_attention_f32(qkv, rs, cos, sin, kc, vc, tmp.ctypes.data, B, H, hd, NS, cbs, pos, max_len,
                                           stream)
        if rc:
            return rc
        t = tmp.reshape(B, NS, H, 4 + hd)
        W = view(Wo, (H, hd, D), (hd * wo_rs, wo_rs, 1), np.float32)
        out = flat(recs, B * NS * H * (4 + D)).reshape(B, NS, H, 4 + D)
        out[..., :2] = t[..., :2]
        live = t[..., 1] > 0
        out[..., 4:] = np.where(live[..., None], np.einsum("bshd,hdn->bshn", np.where(live[..., None], t[..., 4:], 0), W), 0)
        return 0

    def pdn_decode_block_supported(self, D, H, hd, ns):
        return int(hd in (48, 64) and H > 0 and D == H * hd and D <= 1024 and 1 <= ns <= 7 and (ns + 1) * H <= 256)

    def pdn_decode_block_lds_bytes(self, D, H, hd, ns, max_len):
        if ns < 1 or max_len < 1 or not self.pdn_decode_block_supported(D, H, hd, ns):
            return 0
        C = 4 if D % 16 == 0 else (3 if D % 12 == 0 else (2 if D % 8 == 0 else 1))
        SL, G, nqd = 256 // (hd // 4), 256 // (D // 4), D // C // 4
        Go = 256 // nqd
        scf = max(-(-max_len // ns), (SL + 8) * hd, Go * nqd * 4)
        return 4 * (D + max(G * D, 3 * SL * hd) + scf)

    def pdn_decode_block_f32(self, base, base_rs, parts, n_parts, parts_rs, x_out, x_out_rs, norm_w, eps, Wqkv, w_rs, w_bs,
                             cos, sin, kc, vc, cbs, pos, max_len, Wo, wo_rs, recs, B, H, hd, NS, stream):
        D, half = H * hd, hd // 2
        if not self.pdn_decode_block_supported(D, H, hd, NS) or self.pdn_decode_block_lds_bytes(D, H, hd, NS, max_len) > 65536:
            return -2
        if B > 8:
            return -1
        x = np.array(view(base, (B, D), (base_rs, 1), np.float32))
        if n_parts:
            x = (x + view(parts, (B, n_parts, D), (parts_rs, D, 1), np.float32).sum(1)).astype(np.float32)
        view(x_out, (B, D), (x_out_rs, 1), np.float32)[...] = x
        n = (x / np.sqrt((x * x).mean(-1, keepdims=True) + np.float32(eps)) * flat(norm_w, D)).astype(np.float32)
        W = view(Wqkv, (3, D, D), (w_bs, w_rs, 1), np.float32)
        q, k, v = (n @ W[j] for j in range(3))
        p = int(flat(pos, 1, np.int32)[0])
        assert 0 <= p < max_len
        c, s_ = flat(cos + 4 * p * half, half), flat(sin + 4 * p * half, half)

        def rot(t):
            a = np.array(t).reshape(B, H, half, 2)
            out = np.empty_like(a)
            out[..., 0] = a[..., 0] * c - a[..., 1] * s_
            out[..., 1] = a[..., 0] * s_ + a[..., 1] * c
            return out.reshape(B, H, hd)
        Q, Kn, Vn = rot(q), rot(k), np.array(v).reshape(B, H, hd)
        for b in range(B):
            view(kc + 4 * (b * cbs + p * D), (D,), (1,), np.float32)[...] = Kn[b].reshape(D)
            view(vc + 4 * (b * cbs + p * D), (D,), (1,), np.float32)[...] = Vn[b].reshape(D)
        Wov = view(Wo, (H, hd, D), (hd * wo_rs, wo_rs, 1), np.float32)
        out = flat(recs, B * (NS + 1) * H * (4 + D)).reshape(B, NS + 1, H, 4 + D)
        out[...] = 0
        Kc = view(kc, (B, max(p, 1), H, hd), (cbs, D, hd, 1), np.float32)
        Vc = view(vc, (B, max(p, 1), H, hd), (cbs, D, hd, 1), np.float32)
        sc = np.float32(1 / math.sqrt(hd))
        chunk = -(-p // NS)
        for sp in range(NS):
            t0, t1 = sp * chunk, min(p, (sp + 1) * chunk)
            if t0 >= t1:
                out[:, sp, :, 0] = -np.inf
                continue
            ss = np.einsum("bhd,bthd->bht", Q, Kc[:, t0:t1]) * sc
            m = ss.max(-1)
            e = np.exp(ss - m[..., None])
            out[:, sp, :, 0], out[:, sp, :, 1] = m, e.sum(-1)
            out[:, sp, :, 4:] = np.einsum("bhd,hdn->bhn", np.einsum("bht,bthd->bhd", e, Vc[:, t0:t1]), Wov)
        out[:, NS, :, 0] = (Q * Kn).sum(-1) * sc
        out[:, NS, :, 1] = 1.0
        out[:, NS, :, 4:] = np.einsum("bhd,hdn->bhn", Vn, Wov)
        return 0

    @staticmethod
    def _merge_records(rec, ns, H, D):
        """(B, ns, H, 4 + D) softmax partial records -> (B, D) sum over heads of the merged contributions."""
        m, l, o = rec[..., 0], rec[..., 1], rec[..., 4:]
        m = np.where(l > 0, m, -np.inf)
        w = np.where(l > 0, np.exp(m - m.max(1, keepdims=True)), 0).astype(np.float32)
        w = w / (w * l).sum(1, keepdims=True)
        return (w[..., None] * np.where(l[..., None] > 0, o, 0)).sum((1, 2)).astype(np.float32)

    def pdn_decode_mlp_slices(self, F):
        return F // 32 if F > 0 and F % 32 == 0 else 0

    def pdn_decode_mlp_f32(self, base, base_rs, recs, recs_rs, ns, H, x_out, x_out_rs, norm_w, eps, Wg, Wu, w_rs, Wd,
                           wd_rs, parts, parts_rs, B, D, F, stream):
        if B > 8 or D % 4 or D > 1024 or F % 32 or (recs and ns * H > 256):
            return -1
        h = np.array(view(base, (B, D), (base_rs, 1), np.float32))
        if recs:
            h = h + self._merge_records(np.array(view(recs, (B, ns, H, 4 + D), (recs_rs, H * (4 + D), 4 + D, 1),
                                                      np.float32)), ns, H, D)
            if x_out:
                view(x_out, (B, D), (x_out_rs, 1), np.float32)[...] = h
        n = h / np.sqrt((h * h).mean(-1, keepdims=True) + np.float32(eps)) * flat(norm_w, D)
        g = n @ view(Wg, (D, F), (w_rs, 1), np.float32)
        u = n @ view(Wu, (D, F), (w_rs, 1), np.float32)
        a = (g / (np.float32(1) + np.exp(-g)) * u).astype(np.float32)
        Wdv = view(Wd, (F, D), (wd_rs, 1), np.float32)
        J = F // 32
        out = view(parts, (B, J, D), (parts_rs, D, 1), np.float32)
        for j in range(J):
            out[:, j] = a[:, 32 * j:32 * j + 32] @ Wdv[32 * j:32 * j + 32]
        return 0

    def pdn_decode_gemv_sum_f32(self, base, base_rs, parts, n_parts, parts_rs, x_out, x_out_rs, norm_w, eps, W, w_rs,
                                blk_cols, w_bs, bias, y, y_rs, B, K, N, blk_max, blk_arg, stream):
        if K % 4 or K > 1024 or n_parts <= 0:
            return -1
        x = np.array(view(base, (B, K), (base_rs, 1), np.float32))
        x = (x + view(parts, (B, n_parts, K), (parts_rs, K, 1), np.float32).sum(1)).astype(np.float32)
        tmp = np.ascontiguousarray(x)
        if x_out:
            view(x_out, (B, K), (x_out_rs, 1), np.float32)[...] = x
        return self.pdn_decode_gemv_f32(tmp.ctypes.data, K, norm_w, eps, W, w_rs, blk_cols, w_bs, bias, None, 0, y, y_rs,
                                        B, K, N, 0, 0, 0, blk_max, blk_arg, stream)

    def pdn_decode_pick_tick_f32(self, vals, args, B, n, ids, pos, hist, emb, emb_rs, D, x_next, stream):
        v = np.array(flat(vals, B * n).reshape(B, n))
        a = np.array(flat(args, B * n, np.int32).reshape(B, n))
        p = int(flat(pos, 1, np.int32)[0]) if pos else 0
        hrow = flat(int(flat(hist, 1, np.int64)[0]) + 8 * p * B, B, np.int64) if hist else None
        for b in range(B):
            best = v[b].max()
            flat(ids, B, np.int64)[b] = a[b][v[b] == best].min()
            if hrow is not None:
                hrow[b] = flat(ids, B, np.int64)[b]
            if emb:
                tok = int(flat(ids, B, np.int64)[b])
                flat(x_next, B * D).reshape(B, D)[b] = flat(emb + 4 * tok * emb_rs, D)
        if pos:
            flat(pos, 1, np.int32)[0] += 1
        return 0

    def pdn_decode_argmax_tick_f32(self, logits, rs, B, V, ids, pos, stream):
        flat(ids, B, np.int64)[...] = view(logits, (B, V), (rs, 1), np.float32).argmax(-1)
        if pos:
            flat(pos, 1, np.int32)[0] += 1
        return 0

    def pdn_cross_entropy_colsum_workspace_bytes(self, rows, V):
        return 256 * V * 4 if (V >= 4096 and V % 4 == 0 and V <= 32768 and rows > 0) else 0

    def pdn_cross_entropy_fwd_bwd_f32(self, logits, targets, rows, V, mean, gscale, loss_row, lse_row,
                                      loss_out, dlogits, colsum, ws, wsb, err, stream):
        self.pdn_cross_entropy_fwd_f32(logits, targets, rows, V, mean, loss_row, lse_row, loss_out, err, stream)
        rc = self.pdn_cross_entropy_bwd_f32(logits, targets, lse_row, None, gscale, dlogits, rows, V, stream)
        if V <= 32 and rows >= 1024 and not colsum:
            self._count(18)                     # ce_small_kernel: one thread per row
        if colsum:
            flat(colsum, V)[...] = flat(dlogits, rows * V).reshape(rows, V).sum(0)
        return rc

    # -- RNN / LSTM cell pointwise halves -----------------------------------------------------------------
    def pdn_rnn_cell_fwd_f32(self, lin, y, n, act, stream):
        v = np.array(flat(lin, n))
        flat(y, n)[...] = self._tanh(v) if act == 0 else np.maximum(np.float32(0), v)
        return 0

    def pdn_rnn_cell_bwd_f32(self, lin, y, dy, dlin, n, act, stream):
        yy, g = np.array(flat(y, n)), flat(dy, n)
        flat(dlin, n)[...] = (1 - yy * yy) * g if act == 0 else (yy == flat(lin, n)) * g
        return 0

    def pdn_lstm_cell_fwd_f32(self, lin, c, gates, tc, hc, B, H, stream):
        l = np.array(flat(lin, B * 4 * H).reshape(B, 4 * H))
        g = flat(gates, B * 4 * H).reshape(B, 4 * H)
        g[:, :3 * H] = self._sig(l[:, :3 * H])
        g[:, 3 * H:] = self._tanh(l[:, 3 * H:])
        cn = g[:, :H] * flat(c, B * H).reshape(B, H) + g[:, H:2 * H] * g[:, 3 * H:]
        t = self._tanh(cn)
        flat(tc, B * H).reshape(B, H)[...] = t
        out = flat(hc, B * 2 * H).reshape(B, 2 * H)
        out[:, :H], out[:, H:] = g[:, 2 * H:3 * H] * t, cn
        return 0

    def pdn_lstm_cell_bwd_f32(self, dhc, gates, tc, c, dlin, dc_prev, B, H, stream):
        d = flat(dhc, B * 2 * H).reshape(B, 2 * H)
        g = flat(gates, B * 4 * H).reshape(B, 4 * H)
        f, i, o, tg = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        t, cp = flat(tc, B * H).reshape(B, H), flat(c, B * H).reshape(B, H)
        dh = d[:, :H]
        dc = d[:, H:] + dh * o * (1 - t * t)
        dl = flat(dlin, B * 4 * H).reshape(B, 4 * H)
        dl[:, :H] = dc * cp * f * (1 - f)
        dl[:, H:2 * H] = dc * tg * i * (1 - i)
        dl[:, 2 * H:3 * H] = dh * t * o * (1 - o)
        dl[:, 3 * H:] = dc * i * (1 - tg * tg)
        flat(dc_prev, B * H).reshape(B, H)[...] = dc * f
        return 0

    # -- persistent GRU sequence -------------------------------------------------------------------------
    def pdn_gru_seq_supported(self, H): return 1 if H == 32 else 0

    @staticmethod
    def _sig(v):
        out = np.empty_like(v); m = v > 0
        out[m] = 1 / (1 + np.exp(-v[m])); out[~m] = 1 - 1 / (1 + np.exp(v[~m])); return out

    @staticmethod
    def _tanh(v):
        out = np.empty_like(v); m = v > 0
        out[m] = 2 / (1 + np.exp(-2 * v[m])) - 1; out[~m] = 1 - 2 / (1 + np.exp(2 * v[~m])); return out

    def pdn_gru_seq_fwd_f32(self, g1x, g2x, h0, wh1, wh2, z, r, rh, n, out, T, B, H, stream):
        G1, G2 = flat(g1x, T * B * 2 * H).reshape(T, B, 2 * H), flat(g2x, T * B * H).reshape(T, B, H)
        W1, W2 = flat(wh1, H * 2 * H).reshape(H, 2 * H), flat(wh2, H * H).reshape(H, H)
        Z, R, RH, N, O = (flat(p, T * B * H).reshape(T, B, H) for p in (z, r, rh, n, out))
        h = np.array(flat(h0, B * H).reshape(B, H))
        for t in range(T):
            g = self._sig(G1[t] + h @ W1)
            Z[t], R[t] = g[:, :H], g[:, H:]
            RH[t] = R[t] * h
            N[t] = self._tanh(G2[t] + RH[t] @ W2)
            h = (1 - Z[t]) * h + Z[t] * N[t]
            O[t] = h
        return 0

    def pdn_gru_seq_bwd_f32(self, g, z, r, n, out, h0, wh1, wh2, dg1, dg2, dh0, T, B, H, stream):
        G, Z, R, N, O = (flat(p, T * B * H).reshape(T, B, H) for p in (g, z, r, n, out))
        W1, W2 = flat(wh1, H * 2 * H).reshape(H, 2 * H), flat(wh2, H * H).reshape(H, H)
        D1, D2 = flat(dg1, T * B * 2 * H).reshape(T, B, 2 * H), flat(dg2, T * B * H).reshape(T, B, H)
        dh = np.zeros((B, H), np.float32)
        for t in range(T - 1, -1, -1):
            hp_ = O[t - 1] if t > 0 else flat(h0, B * H).reshape(B, H)
            d = dh + G[t]
            D2[t] = (1 - N[t] * N[t]) * (d * Z[t])
            D1[t, :, :H] = Z[t] * (1 - Z[t]) * (d * (N[t] - hp_))
            drh = D2[t] @ W2.T
            D1[t, :, H:] = R[t] * (1 - R[t]) * (drh * hp_)
            dh = d * (1 - Z[t]) + drh * R[t] + D1[t] @ W1.T
        if dh0:
            flat(dh0, B * H).reshape(B, H)[...] = dh
        return 0

    # -- last-axis LayerNorm, gated sigmoid -----------------------------------------------------------
    def pdn_layernorm_bwd_workspace_bytes(self, rows, cols): return 2 * 1024 * cols * 4

    def pdn_layernorm_fwd_f32(self, x, w, b, y, mean, rstd, rows, cols, eps, stream):
        a = np.array(flat(x, rows * cols).reshape(rows, cols))
        mu = a.mean(-1, keepdims=True)
        sd = np.sqrt(np.square(a - mu).mean(-1, keepdims=True) + np.float32(eps))
        flat(mean, rows)[...] = mu[:, 0]
        flat(rstd, rows)[...] = 1 / sd[:, 0]
        flat(y, rows * cols).reshape(rows, cols)[...] = (a - mu) / sd * flat(w, cols) + flat(b, cols)
        return 0

    def pdn_layernorm_bwd_f32(self, x, w, mean, rstd, dy, res, dx, dw, db, acc, rows, cols, ws, wsb, stream):
        a = np.array(flat(x, rows * cols).reshape(rows, cols))
        g = np.array(flat(dy, rows * cols).reshape(rows, cols))
        mu, rs = flat(mean, rows)[:, None], flat(rstd, rows)[:, None]
        xh = (a - mu) * rs
        dz = g * flat(w, cols)
        r = (dz - dz.mean(-1, keepdims=True) - xh * (dz * xh).mean(-1, keepdims=True)) * rs
        if res:
            r = r + flat(res, rows * cols).reshape(rows, cols)
        flat(dx, rows * cols).reshape(rows, cols)[...] = r
        for out, v in ((dw, (g * xh).sum(0)), (db, g.sum(0))):
            if out:
                d = flat(out, cols)
                d[...] = d + v if acc else v
        return 0

    def pdn_gated_sigmoid_fwd_f32(self, x, y, alpha, n, stream):
        v = flat(x, n)
        flat(y, n)[...] = v / (1 + np.exp(-np.float32(alpha) * v))
        return 0

    def pdn_gated_sigmoid_bwd_f32(self, x, dy, dx, alpha, n, stream):
        v, a = np.array(flat(x, n)), np.float32(alpha)
        sg = 1 / (1 + np.exp(-a * v))
        flat(dx, n)[...] = flat(dy, n) * sg * (1 + a * v * (1 - sg))
        return 0

    # -- column-statistics normalisation (reference LayerNorm / BatchNorm1d) -----------------
    def pdn_colnorm_workspace_bytes(self, rows, cols): return ((rows + 255) // 256 * 2 + 2) * cols * 4

    def pdn_colnorm_fwd_f32(self, x, w, b, y, mean, rstd, rmean, rvar, momentum, eps, rows, cols, ws, wsb, stream):
        a = np.array(flat(x, rows * cols).reshape(rows, cols)).astype(np.float64)   # idealised statistics
        mu = a.mean(0)
        var = np.square(a - mu).mean(0)
        rs = 1.0 / np.sqrt(var + eps)
        flat(mean, cols)[...] = mu
        flat(rstd, cols)[...] = rs
        flat(y, rows * cols).reshape(rows, cols)[...] = (a - mu) * rs * flat(w, cols) + flat(b, cols)
        if rmean:
            flat(rmean, cols)[...] = flat(rmean, cols) * np.float32(1 - momentum) + np.float32(momentum) * mu
        if rvar:
            flat(rvar, cols)[...] = flat(rvar, cols) * np.float32(1 - momentum) + np.float32(momentum) * var
        return 0

    def pdn_colnorm_bwd_f32(self, x, w, mean, rstd, dy, dx, dw, db, acc, rows, cols, ws, wsb, stream):
        a = np.array(flat(x, rows * cols).reshape(rows, cols))
        g = np.array(flat(dy, rows * cols).reshape(rows, cols))
        mu, rs = flat(mean, cols), flat(rstd, cols)
        xh = (a - mu) * rs
        sdb, sdw = g.sum(0), (g * xh).sum(0)
        if dx:
            flat(dx, rows * cols).reshape(rows, cols)[...] = flat(w, cols) * rs * (g - sdb / rows - xh * (sdw / rows))
        if dw:
            flat(dw, cols)[...] = flat(dw, cols) + sdw if acc else sdw
        if db:
            flat(db, cols)[...] = flat(db, cols) + sdb if acc else sdb
        return 0

    # -- GRU gate algebra ------------------------------------------------------------------
    @staticmethod
    def _sig(x):
        with np.errstate(over="ignore"):
            return np.where(x > 0, 1 / (1 + np.exp(-x)), 1 - 1 / (1 + np.exp(x))).astype(np.float32)

    @staticmethod
    def _tanh(x):
        with np.errstate(over="ignore"):
            return np.where(x > 0, 2 / (1 + np.exp(-2 * x)) - 1, 1 - 2 / (1 + np.exp(2 * x))).astype(np.float32)

    def pdn_gru_gates_fwd_f32(self, g1, h, z, r, rh, B, H, stream):
        G = flat(g1, B * 2 * H).reshape(B, 2 * H)
        hh = flat(h, B * H).reshape(B, H)
        zz, rr = self._sig(G[:, :H]), self._sig(G[:, H:])
        flat(z, B * H).reshape(B, H)[...] = zz
        flat(r, B * H).reshape(B, H)[...] = rr
        flat(rh, B * H).reshape(B, H)[...] = rr * hh
        return 0

    def pdn_gru_out_fwd_f32(self, g2, z, h, n, hnew, B, H, stream):
        t = self._tanh(flat(g2, B * H))
        zz, hh = flat(z, B * H), flat(h, B * H)
        flat(n, B * H)[...] = t
        flat(hnew, B * H)[...] = (1 - zz) * hh + zz * t
        return 0

    def pdn_gru_out_bwd_f32(self, dhn, z, n, h, dg2, dg1, dh, B, H, stream):
        g, zz, t, hh = (np.array(flat(a, B * H).reshape(B, H)) for a in (dhn, z, n, h))
        flat(dg2, B * H).reshape(B, H)[...] = (1 - t * t) * (g * zz)
        flat(dg1, B * 2 * H).reshape(B, 2 * H)[:, :H] = zz * (1 - zz) * (g * (t - hh))
        flat(dh, B * H).reshape(B, H)[...] = g * (1 - zz)
        return 0

    def pdn_gru_gates_bwd_f32(self, drh, r, h, dg1, dh, B, H, stream):
        d, rr, hh = (np.array(flat(a, B * H).reshape(B, H)) for a in (drh, r, h))
        flat(dg1, B * 2 * H).reshape(B, 2 * H)[:, H:] = rr * (1 - rr) * (d * hh)
        flat(dh, B * H).reshape(B, H)[...] += d * rr
        return 0

    def pdn_scale_by_device_scalar_f32(self, x, n, scalar, stream):
        s = flat(scalar, 1)[0]
        if s != 1.0:
            flat(x, n)[...] *= s
        return 0

    def pdn_adam_multi_f32(self, table, nchunks, step, b1, b2, omb1, omb2, eps, wd, gscale, stream):
        tab = flat(table, nchunks * 5, np.int64).reshape(nchunks, 5)
        f = np.float32
        for p_, g_, m_, v_, n in tab:
            p, g, m, v = flat(p_, n), flat(g_, n), flat(m_, n), flat(v_, n)
            gg = g * f(gscale) + f(wd) * p
            m[...] = m * f(b1) + f(omb1) * gg
            v[...] = v * f(b2) + f(omb2) * (gg * gg)
            p -= f(step) * m / (np.sqrt(v) + f(eps))
        return 0

    # -- conv ------------------------------------------------------------------------------------
    @staticmethod
    def _windows(xp, k, s):
        N, C, H, W = xp.shape
        oh, ow = (H - k) // s + 1, (W - k) // s + 1
        s0, s1, s2, s3 = xp.strides
        return oh, ow, (N, C, k, k, oh, ow), (s0, s1, s2, s3, s2 * s, s3 * s)

    def pdn_im2col2d_f32(self, x, N, C, H, W, k, s, p, col, rows, ones_row, stream):
        xp = np.pad(flat(x, N * C * H * W).reshape(N, C, H, W), [(0, 0), (0, 0), (p, p), (p, p)])
        oh, ow, shape, strides = self._windows(xp, k, s)
        ckk, M = C * k * k, oh * ow
        dst = flat(col, N * rows * M).reshape(N, rows, M)
        dst[:, :ckk] = np.lib.stride_tricks.as_strided(xp, shape, strides).reshape(N, ckk, M)
        dst[:, ckk:] = 0.0
        if ones_row:
            dst[:, ckk] = 1.0
        return 0

    # -- direct convolution (same supported-shape rule as csrc/conv_direct.hip) -------------------------
    def pdn_conv2d_direct_supported(self, C, H, W, O, k, s, p):
        if min(C, O, k, s) <= 0 or p < 0 or H + 2 * p < k or W + 2 * p < k:
            return 0
        lim, mask = 150 * 1024, 0

        def fwd_lds(cin, h, w, cout, pad):
            cp, opad = (cin + 1) // 2 * 2, (cout + 31) // 32 * 32
            return opad <= 64 and 4 * (k * k * cp * opad + cp * (h + 2 * pad) * (w + 2 * pad) + opad) + 64 <= lim
        if fwd_lds(C, H, W, O, p):
            mask |= 1
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        if s == 1 and k - 1 - p >= 0 and fwd_lds(O, oh, ow, C, k - 1 - p):
            mask |= 2
        opad, kcols = (O + 31) // 32 * 32, (C * k * k + 1 + 31) // 32 * 32
        M, img_b = oh * ow, 4 * C * (H + 2 * p) * (W + 2 * p)
        mb = 512 if (img_b + 4 * (O + 1) * (M | 1) > 78 * 1024 and M > 512) else M
        if (opad // 32) * (kcols // 32) <= 16 and max(img_b + 4 * (O + 1) * (mb | 1) + 64, 49152) <= lim:
            mask |= 4
        return mask

    def _conv_cols(self, x, N, C, H, W, k, s, p):
        xp = np.pad(flat(x, N * C * H * W).reshape(N, C, H, W), [(0, 0), (0, 0), (p, p), (p, p)])
        oh, ow, shape, strides = self._windows(xp, k, s)
        return np.lib.stride_tricks.as_strided(xp, shape, strides).reshape(N, C * k * k, oh * ow), oh, ow

    def pdn_conv2d_fwd_f32(self, x, w, bias, y, N, C, H, W, O, k, s, p, stream):
        col, oh, ow = self._conv_cols(x, N, C, H, W, k, s, p)
        out = np.matmul(flat(w, O * C * k * k).reshape(O, -1), col)
        if bias:
            out = out + flat(bias, O).reshape(1, O, 1)
        flat(y, N * O * oh * ow).reshape(N, O, oh * ow)[...] = out
        return 0

    def pdn_conv2d_bwd_data_f32(self, dy, w, dx, N, C, H, W, O, k, s, p, stream):
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        g = flat(dy, N * O * oh * ow).reshape(N, O, oh * ow)
        dcol = np.matmul(flat(w, O * C * k * k).reshape(O, -1).T, g)
        dxp = np.zeros((N, C, H + 2 * p, W + 2 * p), np.float32)
        _, _, shape, strides = self._windows(dxp, k, s)
        np.add.at(np.lib.stride_tricks.as_strided(dxp, shape, strides), (...,), dcol.reshape(shape))
        flat(dx, N * C * H * W).reshape(N, C, H, W)[...] = dxp[:, :, p:p + H, p:p + W]
        return 0

    def pdn_conv2d_bwd_weight_workspace_bytes(self, N, C, H, W, O, k, s, p): return 4096

    def pdn_conv2d_bwd_weight_f32(self, x, dy, dw, db, acc, N, C, H, W, O, k, s, p, ws, wsb, stream):
        col, oh, ow = self._conv_cols(x, N, C, H, W, k, s, p)
        g = flat(dy, N * O * oh * ow).reshape(N, O, oh * ow)
        if dw:
            v = np.matmul(g, col.transpose(0, 2, 1)).sum(0).astype(np.float32).reshape(-1)
            d = flat(dw, O * C * k * k)
            d[...] = d + v if acc else v
        if db:
            v = g.sum((0, 2)).astype(np.float32)
            d = flat(db, O)
            d[...] = d + v if acc else v
        return 0

    # -- conv -> relu -> max_pool(2, 2) as one node ------------------------------------------------------
    def pdn_conv2d_relu_pool_supported(self, C, H, W, O, k, s, p):
        d = self.pdn_conv2d_direct_supported(C, H, W, O, k, s, p)
        if not d & 1 or k != 3 or W % 4 or C * H * W > 16 * 1024:
            return 0
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        if ow not in (8, 16, 32) or oh % 2 or (oh * ow) % 32 or (ow == 32 and oh * ow // 32 < 8):
            return 0
        mask = 1
        if d & 2 and ow % 4 == 0 and O * oh * ow <= 16 * 1024:
            mask |= 2
        M = oh * ow
        mb = 512 if (4 * C * (H + 2 * p) * (W + 2 * p) + 4 * (O + 1) * (M | 1) > 78 * 1024 and M > 512) else M
        if d & 4 and W % 4 == 0 and M % 4 == 0 and mb % 4 == 0 and C * H * W <= 8 * 1024 and O * min(mb, M) <= 16 * 1024 \
                and ow % 4 == 0:
            mask |= 4
        return mask

    @staticmethod
    def _pool_hits(y):
        """pooled, hit words of relu -> 2x2 max-pool on y (..., OH, OW): bit p & 31 of word p >> 5 (p = flat position
        within the (OH, OW) plane) = relu(y) == window max and y >= 0."""
        r = np.maximum(y, 0)
        sh = y.shape[:-2] + (y.shape[-2] // 2, 2, y.shape[-1] // 2, 2)
        rw, yw = r.reshape(sh), y.reshape(sh)
        m = rw.max((-3, -1))
        hit = ((rw == m[..., :, None, :, None]) & (yw >= 0)).reshape(y.shape[:-2] + (-1, 32))
        words = (hit.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
        return m.astype(np.float32), words

    @staticmethod
    def _expand(dp, words, OH, OW):
        bits = ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(dp.shape[:-2] + (OH, OW)).astype(bool)
        up = np.repeat(np.repeat(dp, 2, -2), 2, -1)
        return np.where(bits, up, 0).astype(np.float32)

    @staticmethod
    def _conv_quad(C, H, W, O, k, s, p, which):
        """csrc/conv_quad.hip's dispatch restated: the LeNet shapes of examples/pydynet/mnist.py:82-98 (3x3 / 1 / 1)."""
        if (k, s, p) != (3, 1, 1):
            return False
        shapes = {"fwd": ((20, 16, 16, 50), (3, 32, 32, 20)), "dgrad": ((20, 16, 16, 50),),
                  "wgrad": ((20, 16, 16, 50), (3, 32, 32, 20))}[which]
        return (C, H, W, O) in shapes

    def pdn_conv2d_relu_pool_fwd_f32(self, x, w, bias, pooled, mask, N, C, H, W, O, k, s, p, stream):
        if not self.pdn_conv2d_relu_pool_supported(C, H, W, O, k, s, p) & 1:
            return -2
        if self._conv_quad(C, H, W, O, k, s, p, "fwd"):
            self._count(21)
        col, oh, ow = self._conv_cols(x, N, C, H, W, k, s, p)
        out = np.matmul(flat(w, O * C * k * k).reshape(O, -1), col)
        if bias:
            out = out + flat(bias, O).reshape(1, O, 1)
        m, words = self._pool_hits(out.reshape(N, O, oh, ow).astype(np.float32))
        flat(pooled, m.size).reshape(m.shape)[...] = m
        flat(mask, words.size, np.uint32).reshape(words.shape)[...] = words
        return 0

    def _expanded(self, dp, mask, N, O, oh, ow):
        d = np.array(flat(dp, N * O * oh * ow // 4).reshape(N, O, oh // 2, ow // 2))
        m = np.array(flat(mask, N * O * oh * ow // 32, np.uint32).reshape(N, O, oh * ow // 32))
        return np.ascontiguousarray(self._expand(d, m, oh, ow))

    def pdn_conv2d_relu_pool_bwd_data_f32(self, dp, mask, w, dx, N, C, H, W, O, k, s, p, stream):
        if not self.pdn_conv2d_relu_pool_supported(C, H, W, O, k, s, p) & 2:
            return -2
        if self._conv_quad(C, H, W, O, k, s, p, "dgrad"):
            self._count(22)
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = self._expanded(dp, mask, N, O, oh, ow)
        return self.pdn_conv2d_bwd_data_f32(dy.ctypes.data, w, dx, N, C, H, W, O, k, s, p, stream)

    def pdn_conv2d_relu_pool_bwd_weight_f32(self, x, dp, mask, dw, db, acc, N, C, H, W, O, k, s, p, ws, wsb, stream):
        if not self.pdn_conv2d_relu_pool_supported(C, H, W, O, k, s, p) & 4:
            return -2
        if self._conv_quad(C, H, W, O, k, s, p, "wgrad"):
            self._count(23)
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = self._expanded(dp, mask, N, O, oh, ow)
        return self.pdn_conv2d_bwd_weight_f32(x, dy.ctypes.data, dw, db, acc, N, C, H, W, O, k, s, p, ws, wsb, stream)

    def pdn_pool_mask_expand_f32(self, dp, mask, dy, rows, OH, OW, stream):
        d = np.array(flat(dp, rows * OH * OW // 4).reshape(rows, OH // 2, OW // 2))
        m = np.array(flat(mask, rows * OH * OW // 32, np.uint32))
        bits = ((m[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(rows, OH, OW).astype(bool)
        flat(dy, rows * OH * OW).reshape(rows, OH, OW)[...] = np.where(bits, np.repeat(np.repeat(d, 2, 1), 2, 2), 0)
        return 0

    def pdn_col2im2d_f32(self, dcol, N, C, H, W, k, s, p, dx, rows, stream):
        dxp = np.zeros((N, C, H + 2 * p, W + 2 * p), np.float32)
        oh, ow, shape, strides = self._windows(dxp, k, s)
        src = flat(dcol, N * rows * oh * ow).reshape(N, rows, oh * ow)[:, :C * k * k]
        np.add.at(np.lib.stride_tricks.as_strided(dxp, shape, strides), (...,), src.reshape(shape))
        flat(dx, N * C * H * W).reshape(N, C, H, W)[...] = dxp[:, :, p:p + H, p:p + W]
        return 0

    def pdn_pool2d_fwd_f32(self, x, N, C, H, W, k, s, p, mode, y, stream):
        xp = np.pad(flat(x, N * C * H * W).reshape(N, C, H, W), [(0, 0), (0, 0), (p, p), (p, p)])
        oh, ow, shape, strides = self._windows(xp, k, s)
        win = np.lib.stride_tricks.as_strided(xp, shape, strides)
        flat(y, N * C * oh * ow).reshape(N, C, oh, ow)[...] = win.max((2, 3)) if mode == 0 else win.mean((2, 3))
        return 0

    def pdn_pool2d_bwd_f32(self, x, y, dy, N, C, H, W, k, s, p, mode, dx, stream):
        xp = np.pad(flat(x, N * C * H * W).reshape(N, C, H, W), [(0, 0), (0, 0), (p, p), (p, p)])
        oh, ow, shape, strides = self._windows(xp, k, s)
        win = np.lib.stride_tricks.as_strided(xp, shape, strides)
        yy = flat(y, N * C * oh * ow).reshape(N, C, 1, 1, oh, ow)
        g = flat(dy, N * C * oh * ow).reshape(N, C, 1, 1, oh, ow)
        contrib = (win == yy) * g if mode == 0 else np.broadcast_to(g / np.float32(k * k), shape)
        dxp = np.zeros(xp.shape, np.float32)
        np.add.at(np.lib.stride_tricks.as_strided(dxp, shape, [st for st in dxp.strides[:2]] +
                                                  [dxp.strides[2], dxp.strides[3], dxp.strides[2] * s, dxp.strides[3] * s]),
                  (...,), contrib)
        flat(dx, N * C * H * W).reshape(N, C, H, W)[...] = dxp[:, :, p:p + H, p:p + W]
        return 0


def install(monkeypatch):
    """Point hipnp at the emulated library (host memory); returns the emulator."""
    from pydynet_amd import hipnp
    emu = EmulatedLib()
    monkeypatch.setattr(_lib, "_LIB", emu)
    monkeypatch.setattr(_lib, "is_built", lambda: True)
    monkeypatch.setattr(hipnp, "_ws", {})
    monkeypatch.setattr(hipnp, "_err", {})
    monkeypatch.setattr(hipnp, "_state", {"device": 0, "stream": 0, "streams": {}})
    return emu
