"""im2col / col2im / pooling, the direct convolutions and conv -> relu -> max_pool as one node (incl. the dispatch rule of csrc/conv_quad.hip).
(One part of the TEST-ONLY host emulation of the pdnhip C ABI: see tests/abi_emulator/__init__.py.)"""
import ctypes  # noqa: F401
import math  # noqa: F401

import numpy as np  # noqa: F401

from pydynet_amd import _lib  # noqa: F401
from ._base import _NP, _ints, view, flat  # noqa: F401


class ConvMixin:
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
