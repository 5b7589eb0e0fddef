"""Elementwise, reductions and the fused pointwise nodes (softmax, RMSNorm, SwiGLU, RoPE, cross entropy, embedding, Adam ...).
(One part of the TEST-ONLY host emulation of the pdnhip C ABI: see tests/abi_emulator/__init__.py.)"""
import ctypes  # noqa: F401
import math  # noqa: F401

import numpy as np  # noqa: F401

from pydynet_amd import _lib  # noqa: F401
from ._base import _NP, _ints, view, flat  # noqa: F401


class PointwiseMixin:
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
