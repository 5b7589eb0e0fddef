"""GEMM entry points: pdn_gemm_f32 and the resident / fused-epilogue / mask-epilogue projections, the launch counters.
(One part of the TEST-ONLY host emulation of the pdnhip C ABI: see tests/abi_emulator/__init__.py.)"""
import ctypes  # noqa: F401
import math  # noqa: F401

import numpy as np  # noqa: F401

from pydynet_amd import _lib  # noqa: F401
from ._base import _NP, _ints, view, flat  # noqa: F401


class GemmMixin:
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
