"""The graph-replayable greedy decode step (csrc/decode*.hip).
(One part of the TEST-ONLY host emulation of the pdnhip C ABI: see tests/abi_emulator/__init__.py.)"""
import ctypes  # noqa: F401
import math  # noqa: F401

import numpy as np  # noqa: F401

from pydynet_amd import _lib  # noqa: F401
from ._base import _NP, _ints, view, flat  # noqa: F401


class DecodeMixin:
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
