"""RNN / LSTM / GRU kernels, last-axis LayerNorm, gated sigmoid, column-statistics normalisation.
(One part of the TEST-ONLY host emulation of the pdnhip C ABI: see tests/abi_emulator/__init__.py.)"""
import ctypes  # noqa: F401
import math  # noqa: F401

import numpy as np  # noqa: F401

from pydynet_amd import _lib  # noqa: F401
from ._base import _NP, _ints, view, flat  # noqa: F401


class RecurrentNormMixin:
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
