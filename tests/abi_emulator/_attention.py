"""Fused attention (resident / persistent / head dim 128) and the general streaming attention.
(One part of the TEST-ONLY host emulation of the pdnhip C ABI: see tests/abi_emulator/__init__.py.)"""
import ctypes  # noqa: F401
import math  # noqa: F401

import numpy as np  # noqa: F401

from pydynet_amd import _lib  # noqa: F401
from ._base import _NP, _ints, view, flat  # noqa: F401


class AttentionMixin:
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
