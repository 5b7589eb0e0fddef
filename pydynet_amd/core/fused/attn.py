"""Attention nodes: attention (q, k, v given) and qkv_attention (packed projection + RoPE + attention).
(One module of `pydynet_amd.core.fused`; the package docstring lists the reference chains each node replaces.)"""
from __future__ import annotations

import contextlib
import math
import os

import numpy as np

from .norm import rms_norm
from ..tensor import Tensor, _Operator
from ._common import _hip, _L, _contig, _require_f32, _beside, _is_leaf_f32, _dx_of_shared_input


def _attn_layout(x):
    """(row_stride, batch_stride) of a (B, L, H, hd) device array the attention kernels can read in
    place: unit stride along hd, heads packed (stride hd), 16-byte aligned; else None."""
    B, Lx, H, hd = x.shape
    st = x._strides
    if st[3] != 1 or (H > 1 and st[2] != hd) or st[1] % 4 or (B > 1 and st[0] % 4) or x._ptr % 16:
        return None
    return st[1], (st[0] if B > 1 else 0)


def _attn_mask_args(mask, B, H, Lq, Lk):
    """Pointer and (b, h, q, k) element strides of an additive mask broadcastable to (B, H, Lq, Lk)."""
    if mask is None:
        return None, 0, 0, 0, 0, None
    m = mask
    if m.dtype != np.float32:
        m = m.astype(np.float32)
    shape = (1,) * (4 - m.ndim) + tuple(m.shape)
    if m.ndim > 4 or any(s not in (1, t) for s, t in zip(shape, (B, H, Lq, Lk))):
        raise ValueError(f"attention mask of shape {mask.shape} does not broadcast to {(B, H, Lq, Lk)}")
    m = _contig(m).reshape(shape)
    st = [0 if s == 1 else k for s, k in zip(shape, m._strides)]
    return m._ptr, st[0], st[1], st[2], st[3], m


def _key_bias_of(mask, B, H, Lq, Lk):
    """An additive mask that is a function of (batch, key) only -- the (B, 1, 1, Lk) padding mask of
    examples/pydynet/transformer.py:92-96 -- as a contiguous (b, Lk) float32 device array, b in {1, B}; else None."""
    shape = (1,) * (4 - mask.ndim) + tuple(mask.shape) if mask.ndim <= 4 else None
    if shape is None or shape[1] != 1 or shape[2] != 1 or shape[3] != Lk or shape[0] not in (1, B):
        return None
    m = mask if mask.dtype == np.float32 else mask.astype(np.float32)
    return _contig(m).reshape(shape[0], Lk)


def _attn_kernel(B, H, hd, Lq, Lk, start_pos, mask, layouts):
    """'resident' (K/V of a head held in LDS 256 rows at a time: hd 48 / 64, L <= 1024 -- the benchmark shape is
    one chunk; masks that only depend on the key ride in as a key bias), 'stream' (general kernels: other masks, other
    head dims, lengths that are not multiples of 32 -- zero-padding those up to a tile for the resident kernels was
    measured at CLIP's 50 / 77 positions: 1045 / 976 us against 668 / 645 us, the copies cost more than the kernels
    gain) or None (GEMM + softmax composition).  Returns (kind, key bias or None)."""
    if not attention.use_flash or any(l is None for l in layouts):
        return None, None
    ql, kl, vl = layouts
    if Lq == Lk and start_pos == 0 and attention.use_resident and ql == kl == vl:
        kb = _key_bias_of(mask, B, H, Lq, Lk) if mask is not None else None
        if (mask is None or kb is not None) and _L().query("pdn_attention_supported", Lq, hd):
            return "resident", kb
    if _L().query("pdn_attention_stream_supported", hd) and kl == vl:
        return "stream", None
    return None, None


class attention(_Operator):
    """softmax(q k^T / sqrt(hd) + causal_mask + mask) v  per (batch, head).

    q: (B, L, H, hd); k, v: (B, Lk, H, hd) -- the layout the Q/K/V projections produce, consumed
    through strides (no transposes, no copies; views into a packed QKV projection or a KV cache are
    fine).  Output (B, L, H, hd).  `causal` applies the additive -inf upper-triangular mask of
    llm/llama/model.py:199-203 with `start_pos`; `mask` is an optional constant additive mask
    broadcastable to (B, H, L, Lk) (padding masks, llm/clip's causal mask tensor)."""

    use_flash = True      # class switch: False forces the GEMM + softmax path (A/B and tests)
    use_resident = True   # class switch: False sends the benchmark shape through the streaming kernels too

    def __init__(self, q, k, v, causal=True, start_pos=0, mask=None):
        self.causal, self.start_pos = bool(causal), int(start_pos)
        self._mask = mask.data if isinstance(mask, Tensor) else mask
        self._kind = None
        super().__init__(q, k, v)

    def _np_mask(self, Lq, Lk, dtype):
        add = None
        if self.causal and Lq > 1:
            m = np.triu(np.full((Lq, Lq), float("-inf")), k=1)
            add = np.concatenate([np.zeros((Lq, self.start_pos)), m], axis=1).astype(dtype)
        if self._mask is not None:
            mk = np.asarray(self._mask, dtype=dtype)
            add = mk if add is None else add + mk
        return add

    def forward_(self, q, k, v):
        B, Lq, H, hd = q.shape
        Lk = k.shape[1]
        if self.xp is np:
            s = np.matmul(q.data.transpose(0, 2, 1, 3), k.data.transpose(0, 2, 3, 1)) / np.asarray(math.sqrt(hd), q.dtype)
            add = self._np_mask(Lq, Lk, q.dtype)
            if add is not None:
                s = s + add
            e = np.exp(s - s.max(-1, keepdims=True))
            self._p = e / e.sum(-1, keepdims=True)
            return np.ascontiguousarray(np.matmul(self._p, v.data.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3))
        _require_f32(self, q, k, v)
        hp, L = _hip(), _L()
        causal = 1 if (self.causal and Lq > 1) else 0
        layouts = (_attn_layout(q.data), _attn_layout(k.data), _attn_layout(v.data))
        mask_dev = hp.asarray(self._mask) if self._mask is not None else None
        self._kind, kb = _attn_kernel(B, H, hd, Lq, Lk, self.start_pos, mask_dev, layouts)
        if self._kind == "resident":
            # scores stay in registers: one kernel, nothing of size L x L in HBM; lse kept for backward
            qd, kd, vd = q.data, k.data, v.data
            rs, bs = layouts[0]
            out = hp.empty((B, Lq, H, hd), np.float32)
            self._lse = hp.empty((B, H, Lq), np.float32)
            self._res = (qd, kd, vd, kb, rs, bs)
            if kb is None:
                L.call("pdn_attention_fwd_f32", qd._ptr, kd._ptr, vd._ptr, out._ptr, self._lse._ptr,
                       B, H, Lq, hd, rs, bs, H * hd, Lq * H * hd, causal, None, None, hp.stream())
            else:
                # a mask that depends on (batch, key) only: one more rank-1 step of the score product inside the kernels
                L.call("pdn_attention_fwd_bias_f32", qd._ptr, kd._ptr, vd._ptr, out._ptr, self._lse._ptr,
                       B, H, Lq, hd, rs, bs, H * hd, Lq * H * hd, causal, kb._ptr, Lq if kb.shape[0] > 1 else 0,
                       hp.stream())
            return out
        if self._kind == "stream":
            out = hp.empty((B, Lq, H, hd), np.float32)
            self._lse = hp.empty((B, H, Lq), np.float32)
            mp, sb, sh, sq, sk, self._mask_dev = _attn_mask_args(mask_dev, B, H, Lq, Lk)
            # the output is written with the QUERY strides: give the kernel a q-shaped contiguous view
            if layouts[0] != (H * hd, Lq * H * hd if B > 1 else 0):
                self._q_used = q.data.copy()
                layouts = (_attn_layout(self._q_used), layouts[1], layouts[2])
            else:
                self._q_used = q.data
            self._lay = layouts
            L.call("pdn_attention_stream_fwd_f32", self._q_used._ptr, k.data._ptr, v.data._ptr, out._ptr,
                   self._lse._ptr, B, H, Lq, Lk, hd, layouts[0][0], layouts[0][1], layouts[1][0], layouts[1][1],
                   causal, self.start_pos, mp, sb, sh, sq, sk, None, None, hp.stream())
            return out
        p = hp.empty((B, H, Lq, Lk), np.float32)
        hp.gemm(q.data.transpose(0, 2, 1, 3), k.data.transpose(0, 2, 3, 1), p)
        div = math.sqrt(hd)
        if self._mask is not None:
            p = p / np.float32(div) + hp.asarray(self._mask).astype(np.float32)
            div = 1.0
        L.call("pdn_softmax_fwd_f32", p._ptr, p._ptr, B * H * Lq, Lk, div,
               Lq if causal else 0, self.start_pos, hp.stream())
        self._p = p
        out = hp.empty((B, Lq, H, hd), np.float32)
        hp.gemm(p, v.data.transpose(0, 2, 1, 3), out.transpose(0, 2, 1, 3))
        return out

    def backward_all(self, do):
        q, k, v = self.last
        B, Lq, H, hd = q.shape
        Lk = k.shape[1]
        causal = 1 if (self.causal and Lq > 1) else 0
        if self.xp is not np and self._kind == "resident":
            hp, L = _hip(), _L()
            qd, kd, vd, kb, rs, bs = self._res
            self._res = None
            do = _contig(do)
            dq, dk, dv = (hp.empty((B, Lq, H, hd), np.float32) for _ in range(3))
            if rs != H * hd or (B > 1 and bs != Lq * H * hd):
                # gradients are written with the operand strides: strided views (a packed q | k | v projection) get
                # contiguous copies of the operands here
                qd, kd, vd = qd.copy(), kd.copy(), vd.copy()
                rs, bs = H * hd, Lq * H * hd
            ws, wsb = hp.workspace(L.query("pdn_attention_bwd_workspace_bytes", B, H, Lq))
            if kb is None:
                L.call("pdn_attention_bwd_f32", qd._ptr, kd._ptr, vd._ptr, self.data._ptr, do._ptr,
                       self._lse._ptr, dq._ptr, dk._ptr, dv._ptr, B, H, Lq, hd, rs, bs, H * hd, Lq * H * hd,
                       causal, None, None, ws, wsb, hp.stream())
            else:
                L.call("pdn_attention_bwd_bias_f32", qd._ptr, kd._ptr, vd._ptr, self.data._ptr, do._ptr,
                       self._lse._ptr, dq._ptr, dk._ptr, dv._ptr, B, H, Lq, hd, rs, bs, H * hd, Lq * H * hd,
                       causal, kb._ptr, Lq if kb.shape[0] > 1 else 0, ws, wsb, hp.stream())
            return [dq, dk, dv]
        if self.xp is not np and self._kind == "stream":
            hp, L = _hip(), _L()
            do = _contig(do)
            dq = hp.empty(q.shape, np.float32)
            # dk / dv are written with the key strides: contiguous gradients need contiguous k / v
            ksrc, vsrc = _contig(k.data), _contig(v.data)
            dk, dv = hp.empty(k.shape, np.float32), hp.empty(v.shape, np.float32)
            klay = (H * hd, Lk * H * hd if B > 1 else 0)
            mp, sb, sh, sq, sk, keep = _attn_mask_args(self._mask_dev, B, H, Lq, Lk) \
                if self._mask is not None else (None, 0, 0, 0, 0, None)
            ws, wsb = hp.workspace(L.query("pdn_attention_stream_bwd_workspace_bytes", B, H, Lq))
            qlay = (H * hd, Lq * H * hd if B > 1 else 0)
            qsrc = self._q_used
            L.call("pdn_attention_stream_bwd_f32", qsrc._ptr, ksrc._ptr, vsrc._ptr, self.data._ptr, do._ptr,
                   self._lse._ptr, dq._ptr, dk._ptr, dv._ptr, B, H, Lq, Lk, hd, qlay[0], qlay[1], klay[0], klay[1],
                   causal, self.start_pos, mp, sb, sh, sq, sk, None, None, ws, wsb, hp.stream())
            return [dq, dk, dv]
        p = self._p
        if self.xp is np:
            doT = do.transpose(0, 2, 1, 3)
            dv = np.matmul(p.swapaxes(-1, -2), doT).transpose(0, 2, 1, 3)
            dp = np.matmul(doT, v.data.transpose(0, 2, 3, 1))
            ds = (dp - (dp * p).sum(-1, keepdims=True)) * p / np.asarray(math.sqrt(hd), q.dtype)
            dq = np.matmul(ds, k.data.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)
            dk = np.matmul(ds.swapaxes(-1, -2), q.data.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)
            return [dq, dk, dv]
        hp, L = _hip(), _L()
        doT = do.transpose(0, 2, 1, 3)
        dv = hp.empty(v.shape, np.float32)
        hp.gemm(p.swapaxes(-1, -2), doT, dv.transpose(0, 2, 1, 3))                  # P^T dO
        dp = hp.empty(p.shape, np.float32)
        hp.gemm(doT, v.data.transpose(0, 2, 3, 1), dp)                              # dO V^T
        L.call("pdn_softmax_bwd_f32", p._ptr, dp._ptr, dp._ptr, B * H * Lq, Lk, math.sqrt(hd), hp.stream())
        dq, dk = hp.empty(q.shape, np.float32), hp.empty(k.shape, np.float32)
        hp.gemm(dp, k.data.transpose(0, 2, 1, 3), dq.transpose(0, 2, 1, 3))         # dS K
        hp.gemm(dp.swapaxes(-1, -2), q.data.transpose(0, 2, 1, 3), dk.transpose(0, 2, 1, 3))  # dS^T Q
        return [dq, dk, dv]


class qkv_attention(_Operator):
    """Training-path self-attention front end as ONE tape node (llm/llama/model.py:92-121):
    the three bias-free projections write the column blocks of ONE packed (tokens, 3 * dim) buffer
    (a single batched GEMM when the weights are equally spaced in memory, as `Attention.move` packs
    them), the fused causal attention reads q / k / v from it through strides with RoPE applied inside
    its kernels.  Backward: attention backward into one packed (tokens, 3 * dim) buffer (dq, dk already
    rotated back), the three weight gradients as ONE batched wave-streaming GEMM (when the leaf
    gradients are equally spaced, e.g. in the flat gradient buffer) and dx as ONE GEMM that contracts
    over all 3 * dim columns against the column-packed weights.  x: (B, L, D); returns (B, L, H, hd)."""

    folds_existing = True
    enabled = True          # class switch: False sends Attention through the separate nodes (tests, A/B)
    rope_epilogue = os.environ.get("PDN_NO_ROPE_EPILOGUE", "0") != "1"   # RoPE in the store of the q | k | v projection
    rope_min_rows = 4096
    prerotate = os.environ.get("PDN_NO_PREROTATE", "0") != "1"           # (same-box A/B switch)
    _rope_tables = {}       # (cos ptr, sin ptr, L, hd) -> expanded (L, hd, 2) table for the projection's epilogue

    @staticmethod
    def _rope_table(cos, sin, Lq, hd):
        hp, L = _hip(), _L()
        key = (cos._ptr, sin._ptr, Lq, hd)
        ent = qkv_attention._rope_tables.get(key)
        if ent is None:
            if len(qkv_attention._rope_tables) > 16:
                qkv_attention._rope_tables.clear()
            tab = hp.empty((Lq, hd, 2), np.float32)
            L.call("pdn_rope_table_f32", cos._ptr, sin._ptr, tab._ptr, Lq, hd, hp.stream())
            # (the tables are kept alive with the entry: the key is made of their addresses)
            ent = qkv_attention._rope_tables[key] = (tab, cos, sin)
        return ent[0]

    def __init__(self, x, wq, wk, wv, cos, sin, n_heads):
        self._cos, self._sin, self.H = cos, sin, int(n_heads)
        super().__init__(x, wq, wk, wv)

    @staticmethod
    def _resident(L, hd):
        return bool(attention.use_resident and _L().query("pdn_attention_supported", L, hd))

    @staticmethod
    def applicable(x, L, hd):
        if not (qkv_attention.enabled and attention.use_flash and x.device.is_hip and x.dtype == np.float32
                and x.ndim == 3):
            return False
        return qkv_attention._resident(L, hd) or bool(_L().query("pdn_attention_stream_supported", hd))

    @staticmethod
    def _blocks(buf, T, D):
        """The three (T, D) column blocks of a packed (T, 3D) buffer as one (3, T, D) strided view."""
        hp = _hip()
        return hp.ndarray(buf._buf, buf._ptr, (3, T, D), (D, 3 * D, 1), buf.dtype)

    def forward_(self, x, wq, wk, wv):
        _require_f32(self, x, wq, wk, wv, self._cos, self._sin)
        hp, L = _hip(), _L()
        B, Lq, D = x.shape
        H, hd, T = self.H, D // self.H, B * Lq
        qkv = hp.empty((T, 3 * D), np.float32)
        blocks = self._blocks(qkv, T, D)
        ws = [_contig(w.data) for w in (wq, wk, wv)]
        stack = hp.stacked_view(ws)
        cos, sin = _contig(self._cos.data), _contig(self._sin.data)
        resident = qkv_attention._resident(Lq, hd)
        # RoPE in the projection's store (q, k leave rotated; the attention kernels read them as they are and only
        # rotate dq, dk back), or -- shapes that kernel does not take -- inside the attention kernels' loads
        self.rotated = bool(qkv_attention.rope_epilogue and resident and stack is not None
                            and abs(stack._strides[0]) < (1 << 40)
                            and T >= qkv_attention.rope_min_rows
                            and L.query("pdn_qkv_rope_supported", T, D, D, Lq, hd))
        # a still-deferred RMSNorm in front (fused.rms_norm): its rows are normalised in this projection's A load
        self.norm_folded = bool(self.rotated and isinstance(x, rms_norm) and x._pending is not None
                                and L.query("pdn_qkv_rope_norm_supported", T, D, D, Lq, hd))
        if self.norm_folded:
            raw_t, wn = x._pending
            raw = _contig(raw_t.data)
            xn, rms = hp.empty(x.shape, np.float32), hp.empty((T,), np.float32)
            tab = self._rope_table(cos, sin, Lq, hd)
            L.call("pdn_qkv_rope_norm_fwd_f32", raw._ptr, _contig(wn.data)._ptr, x.eps, xn._ptr, rms._ptr, ws[0]._ptr,
                   (ws[1]._ptr - ws[0]._ptr) // 4, qkv._ptr, tab._ptr, T, D, D, Lq, hd, D, hp.stream())
            x._adopt(raw, rms, xn)
        x2 = _contig(x.data).reshape(T, D)
        if self.norm_folded:
            pass
        elif self.rotated:
            tab = self._rope_table(cos, sin, Lq, hd)
            L.call("pdn_qkv_rope_fwd_f32", x2._ptr, ws[0]._ptr, (ws[1]._ptr - ws[0]._ptr) // 4, qkv._ptr, tab._ptr,
                   T, D, D, Lq, hd, D, hp.stream())
        elif stack is not None:
            hp.gemm(x2, stack, blocks)
        else:
            for i in range(3):
                hp.gemm(x2, ws[i], blocks[i])
        # widths whose projection has no RoPE store (contraction other than 288): q | k rotated IN PLACE in the packed buffer
        # (one pass over two thirds of it), so that the attention runs on the persistent kernels -- 2.4 instead of 3.3-4.3 ns
        # per tile pair forward -- which take rotation-free operands only; the backward is the rotated one either way
        self.prerotated = bool(not self.rotated and resident and qkv_attention.rope_epilogue and qkv_attention.prerotate
                               and T >= qkv_attention.rope_min_rows
                               and L.query("pdn_attention_persistent_supported", Lq, hd))
        if self.prerotated:
            L.call("pdn_rope_rows_f32", qkv._ptr, cos._ptr, sin._ptr, qkv._ptr, T, Lq, 2 * H, hd, 3 * D, 3 * D, 0, hp.stream())
            self.rotated = True
        out = hp.empty((B, Lq, H, hd), np.float32)
        lse = hp.empty((B, H, Lq), np.float32)
        q, k, v = qkv._ptr, qkv._ptr + 4 * D, qkv._ptr + 8 * D
        if resident:
            L.call("pdn_attention_fwd_f32", q, k, v, out._ptr, lse._ptr, B, H, Lq, hd, 3 * D, Lq * 3 * D,
                   D, Lq * D, 1, None if self.rotated else cos._ptr, None if self.rotated else sin._ptr, hp.stream())
        else:                   # any length / head dim: key tiles stream through LDS, RoPE still in the loads
            if (3 * D) % 4 or D % 4:
                raise ValueError("qkv_attention: dim must be a multiple of 4")
            # (the streaming kernels write o with the query strides: give them a dense q copy)
            qd = blocks[0].copy()
            L.call("pdn_attention_stream_fwd_f32", qd._ptr, k, v, out._ptr, lse._ptr,
                   B, H, Lq, Lq, hd, D, Lq * D, 3 * D, Lq * 3 * D, 1 if Lq > 1 else 0, 0, None, 0, 0, 0, 0,
                   cos._ptr, sin._ptr, hp.stream())
        self._saved = (x2, qkv, lse, cos, sin)
        return out

    def backward_all(self, do):
        hp, L = _hip(), _L()
        x, wq, wk, wv = self.last
        B, Lq, D = x.shape
        H, hd, T = self.H, D // self.H, B * Lq
        x2, qkv, lse, cos, sin = self._saved
        do = _contig(do)
        dqkv = hp.empty((T, 3 * D), np.float32)
        dblocks = self._blocks(dqkv, T, D)
        q, k, v = qkv._ptr, qkv._ptr + 4 * D, qkv._ptr + 8 * D
        dq, dk, dv = dqkv._ptr, dqkv._ptr + 4 * D, dqkv._ptr + 8 * D
        if qkv_attention._resident(Lq, hd):
            ws_, wsb = hp.workspace(L.query("pdn_attention_bwd_workspace_bytes", B, H, Lq))
            L.call("pdn_attention_bwd_rotated_f32" if self.rotated else "pdn_attention_bwd_f32", q, k, v, self.data._ptr,
                   do._ptr, lse._ptr, dq, dk, dv, B, H, Lq, hd,
                   3 * D, Lq * 3 * D, D, Lq * D, 1, cos._ptr, sin._ptr, ws_, wsb, hp.stream())
        else:
            ws_, wsb = hp.workspace(L.query("pdn_attention_stream_bwd_workspace_bytes", B, H, Lq))
            qd, dqd = self._blocks(qkv, T, D)[0].copy(), hp.empty((T, D), np.float32)
            L.call("pdn_attention_stream_bwd_f32", qd._ptr, k, v, self.data._ptr, do._ptr, lse._ptr, dqd._ptr, dk, dv,
                   B, H, Lq, Lq, hd, D, Lq * D, 3 * D, Lq * 3 * D, 1 if Lq > 1 else 0, 0, None, 0, 0, 0, 0,
                   cos._ptr, sin._ptr, ws_, wsb, hp.stream())
            dblocks[0] = dqd
        grads = [None] * 4
        weights = (wq, wk, wv)
        gstack = None
        if all(w.requires_grad and _is_leaf_f32(w) for w in weights):
            gstack = hp.stacked_view([w.grad for w in weights])
        side = _beside(hp, D, 3 * D, x.requires_grad and any(w.requires_grad for w in weights))
        dws = [hp.empty(w.shape, np.float32) if gstack is None and w.requires_grad and not _is_leaf_f32(w) else None
               for w in weights]
        with side or contextlib.nullcontext():
            if gstack is not None:
                hp.gemm(x2.T, dblocks, gstack, beta=1.0)              # three x^T @ d_i in one launch
            else:
                for i, w in enumerate(weights):
                    if not w.requires_grad:
                        continue
                    if dws[i] is None:
                        hp.gemm(x2.T, dblocks[i], w.grad, beta=1.0)
                    else:
                        hp.gemm(x2.T, dblocks[i], dws[i])
                        grads[1 + i] = dws[i]
        if x.requires_grad:
            grads[0] = _dx_of_shared_input(hp, self, x, dqkv, (wq, wk, wv), T, D)
        if side is not None:
            side.join()
        return grads
