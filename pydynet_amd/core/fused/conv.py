"""Convolution nodes: conv2d (direct / im2col), the deferred relu, conv2d_relu_pool, pool2d.
(One module of `pydynet_amd.core.fused`; the package docstring lists the reference chains each node replaces.)"""
from __future__ import annotations

import numpy as np

from ..tensor import _Operator
from ._common import _hip, _L, _contig, hip_f32, _require_f32, _is_leaf_f32, _Deferred


class relu(_Deferred, _Operator):
    """maximum(0., x); the gradient passes where out == x, i.e. also at x == 0 (reference quirk).
    relu of a still-deferred conv2d node is deferred as well (see _Deferred)."""

    def __init__(self, x):
        if isinstance(x, conv2d) and x._pending is not None:
            self._init_deferred((x,), x.shape, x.dtype)
        else:
            super().__init__(x)

    def forward_(self, x):
        return self.xp.maximum(np.array(0., dtype=x.dtype) if self.xp is np else 0.0, x.data)

    def backward_all(self, dy):
        x = self.last[0]
        if self.xp is np or x.dtype != np.float32:
            return [(self.data == x.data) * dy]
        hp, L = _hip(), _L()
        xd, dy = _contig(x.data), _contig(dy)
        dx = hp.empty(x.shape, np.float32)
        L.call("pdn_relu_bwd_f32", xd._ptr, dy._ptr, dx._ptr, dy.size, hp.stream())
        return [dx]


class conv2d(_Deferred, _Operator):
    """Square-kernel 2-D convolution (nn/functional.py:254-281).

    HIP device, LeNet-class shapes (the padded image and the weights fit in LDS): direct
    implicit-GEMM kernels -- nothing of the im2col buffer ever exists in HBM (`pdn_conv2d_*`).
    Other shapes: im2col + ONE batched GEMM per direction: the im2col buffer keeps the reference
    layout (N, C, kh, kw, oh, ow) in its first C*k*k rows and pads the contraction to a multiple of
    4; per image the packed weight (O, Kp) multiplies it into a contiguous NCHW output (the reference
    returns an NHWC buffer viewed as NCHW: same values); the bias (1, O, 1, 1) is column K of the
    packed weight against a row of ones, so `+ bias` and its gradient ride inside the GEMMs.
    `node._col` is the reference-layout im2col buffer (formed on demand on the direct path)."""

    use_direct = True       # class switch: False forces the im2col + GEMM path (tests, A/B)
    defer = True            # class switch: False runs the kernel at construction (no conv + relu + pool fusion)

    def __init__(self, x, kernel, bias=None, padding=0, stride=1):
        self.padding, self.stride = int(padding), int(stride)
        self.has_bias = bias is not None
        inputs = (x, kernel, bias) if self.has_bias else (x, kernel)
        if conv2d.defer and type(self) is conv2d and self._fusable(x, kernel, bias):
            N, C, H, W, O, k, oh, ow = self._dims(x, kernel)
            self._init_deferred(inputs, (N, O, oh, ow), np.float32)
        else:
            super().__init__(*inputs)

    def _fusable(self, x, kernel, bias):
        """A shape / device the fused conv + relu + 2x2 max-pool kernel takes (then the node is deferred)."""
        if not (conv2d.use_direct and x.device.is_hip and hip_f32(x, kernel, bias)) or x.ndim != 4 or kernel.ndim != 4:
            return False
        N, C, H, W, O, k, oh, ow = self._dims(x, kernel)
        if kernel.shape[1] != C or kernel.shape[3] != k or (bias is not None and bias.size != O):
            return False
        return bool(_L().query("pdn_conv2d_relu_pool_supported", C, H, W, O, k, self.stride, self.padding) & 1)

    def _dims(self, x, kernel):
        N, C, H, W = x.shape
        O, _, k, _ = kernel.shape
        oh = (H + 2 * self.padding - k) // self.stride + 1
        ow = (W + 2 * self.padding - k) // self.stride + 1
        return N, C, H, W, O, k, oh, ow

    def _im2col_np(self, xd, k):
        p, s = self.padding, self.stride
        xp_ = np.pad(xd, [(0, 0), (0, 0), (p, p), (p, p)], "constant")
        N, C, H, W = xp_.shape
        oh, ow = (H - k) // s + 1, (W - k) // s + 1
        s0, s1, s2, s3 = xp_.strides
        return np.lib.stride_tricks.as_strided(xp_, (N, C, k, k, oh, ow), (s0, s1, s2, s3, s2 * s, s3 * s)).copy()

    def forward_(self, x, kernel, bias=None):
        N, C, H, W, O, k, oh, ow = self._dims(x, kernel)
        if self.xp is np:
            self._col_np = self._im2col_np(x.data, k)
            a = self._col_np.transpose(0, 4, 5, 1, 2, 3).reshape(N * oh * ow, -1)
            out = a @ kernel.data.reshape(O, -1).T
            if bias is not None:
                out = out + bias.data.reshape(1, O)
            return out.reshape(N, oh, ow, O).transpose(0, 3, 1, 2)
        _require_f32(self, x, kernel, bias)
        hp, L = _hip(), _L()
        self._xd = _contig(x.data)
        self._k_shape = tuple(kernel.shape)
        self._kernel_data = kernel.data
        self._bias_data = bias.data if bias is not None else None
        self._colp = self._wp = None
        self._direct = L.query("pdn_conv2d_direct_supported", C, H, W, O, k, self.stride, self.padding) \
            if conv2d.use_direct else 0
        out = hp.empty((N, O, oh, ow), np.float32)                 # NCHW, contiguous
        if self._direct & 1:
            wd = _contig(kernel.data)
            L.call("pdn_conv2d_fwd_f32", self._xd._ptr, wd._ptr,
                   _contig(bias.data)._ptr if bias is not None else None, out._ptr, N, C, H, W, O, k,
                   self.stride, self.padding, hp.stream())
            return out
        colp, wp = self._ensure_col(), self._ensure_wp()
        hp.gemm(wp, colp, out.reshape(N, O, oh * ow))              # per image (O,Kp) @ (Kp,M)
        return out

    # -- explicit im2col operands (generic path, and the bit-exact `col` of the parity tests) ------
    def _ensure_col(self):
        if getattr(self, "_colp", None) is None:
            hp, L = _hip(), _L()
            N, C, H, W = self._xd.shape
            k = self._k_shape[2]
            M = ((H + 2 * self.padding - k) // self.stride + 1) * ((W + 2 * self.padding - k) // self.stride + 1)
            K = C * k * k
            # contraction padded to a multiple of 4 (16-byte GEMM path); with a bias, row K of the
            # im2col buffer is ones and column K of the packed weight is the bias
            self._Kp = (K + (1 if self.has_bias else 0) + 3) // 4 * 4
            self._colp = hp.empty((N, self._Kp, M), np.float32)
            L.call("pdn_im2col2d_f32", self._xd._ptr, N, C, H, W, k, self.stride, self.padding, self._colp._ptr,
                   self._Kp, 1 if self.has_bias else 0, hp.stream())
        return self._colp

    def _ensure_wp(self):
        if getattr(self, "_wp", None) is None:
            hp = _hip()
            self._ensure_col()
            O, C, k, _ = self._k_shape
            K = C * k * k
            wp = hp.zeros((O, self._Kp), np.float32)
            wp[:, :K] = self._kernel_data.reshape(O, K)
            if self.has_bias:
                wp[:, K] = self._bias_data.reshape(O)
            self._wp = wp
        return self._wp

    @property
    def _col(self):
        """The im2col buffer in the reference layout (N, C, kh, kw, oh, ow) (a view on the HIP path)."""
        self.data                                        # (a deferred node runs its kernel now)
        if self.xp is np:
            return self._col_np
        N, C, H, W = self._xd.shape                      # (the node's edges are gone after backward)
        k = self._k_shape[2]
        oh = (H + 2 * self.padding - k) // self.stride + 1
        ow = (W + 2 * self.padding - k) // self.stride + 1
        return self._ensure_col()[:, :C * k * k].reshape(N, C, k, k, oh, ow)

    def backward_all(self, g):
        x, kernel = self.last[0], self.last[1]
        bias = self.last[2] if self.has_bias else None
        N, C, H, W, O, k, oh, ow = self._dims(x, kernel)
        K, M = C * k * k, oh * ow
        grads = [None] * len(self.last)
        if self.xp is np:
            g2 = g.transpose(0, 2, 3, 1).reshape(N * M, O)
            a = self._col_np.transpose(0, 4, 5, 1, 2, 3).reshape(N * M, K)
            if kernel.requires_grad:
                grads[1] = (g2.T @ a).reshape(kernel.shape)
            if bias is not None and bias.requires_grad:
                grads[2] = g2.sum(0).reshape(bias.shape)
            if x.requires_grad:
                dcol = (g2 @ kernel.data.reshape(O, K)).reshape(N, oh, ow, C, k, k).transpose(0, 3, 4, 5, 1, 2)
                p, s = self.padding, self.stride
                dxp = np.zeros((N, C, H + 2 * p, W + 2 * p), dtype=g.dtype)
                s0, s1, s2, s3 = dxp.strides
                view = np.lib.stride_tricks.as_strided(dxp, (N, C, k, k, oh, ow), (s0, s1, s2, s3, s2 * s, s3 * s))
                np.add.at(view, (...,), dcol)
                grads[0] = dxp[:, :, p:p + H, p:p + W] if p else dxp
            return grads
        hp, L = _hip(), _L()
        gc = _contig(g)
        need_dw = kernel.requires_grad
        need_db = bias is not None and bias.requires_grad
        if (need_dw or need_db) and self._direct & 4:
            # dW and db straight into the leaves' gradient buffers when they are float32 leaves
            direct_w = need_dw and _is_leaf_f32(kernel)
            direct_b = need_db and _is_leaf_f32(bias)
            if need_dw and need_db and direct_w != direct_b:
                direct_w = direct_b = False              # one accumulate flag: keep both on the same side
            dw = (kernel.grad if direct_w else hp.empty(kernel.shape, np.float32)) if need_dw else None
            db = (bias.grad if direct_b else hp.empty((O,), np.float32)) if need_db else None
            ws, wsb = hp.workspace(L.query("pdn_conv2d_bwd_weight_workspace_bytes", N, C, H, W, O, k,
                                           self.stride, self.padding))
            L.call("pdn_conv2d_bwd_weight_f32", self._xd._ptr, gc._ptr, dw._ptr if dw is not None else None,
                   db._ptr if db is not None else None, 1 if (direct_w or direct_b) else 0, N, C, H, W, O, k,
                   self.stride, self.padding, ws, wsb, hp.stream())
            if need_dw and not direct_w:
                grads[1] = dw
            if need_db and not direct_b:
                grads[2] = db.reshape(bias.shape)
        elif need_dw or need_db:
            colp = self._ensure_col()
            Kp = self._Kp
            # per image g (O,M) @ col^T (M,Kp); column K of the sum is the bias gradient
            part = hp.empty((N, O, Kp), np.float32)
            hp.gemm(gc.reshape(N, O, M), colp.transpose(0, 2, 1), part)
            dwp = part.sum(0)
            if need_dw:
                grads[1] = dwp[:, :K].reshape(kernel.shape)
            if need_db:
                grads[2] = dwp[:, K].reshape(bias.shape)
        if x.requires_grad:
            dx = hp.empty((N, C, H, W), np.float32)
            if self._direct & 2:
                L.call("pdn_conv2d_bwd_data_f32", gc._ptr, _contig(kernel.data)._ptr, dx._ptr, N, C, H, W, O, k,
                       self.stride, self.padding, hp.stream())
            else:
                wp = self._ensure_wp()
                Kp = self._Kp
                dcol = hp.empty((N, Kp, M), np.float32)
                hp.gemm(wp.T, gc.reshape(N, O, M), dcol)                       # (Kp,O) @ (O,M) per image
                L.call("pdn_col2im2d_f32", dcol._ptr, N, C, H, W, k, self.stride, self.padding, dx._ptr, Kp,
                       hp.stream())
            grads[0] = dx
        return grads


class conv2d_relu_pool(conv2d):
    """max_pool2d(relu(conv2d(x, w) + b), 2, 2) as ONE node (mnist.py:92-95; functional.py:31-32, 254-339).

    Forward: the direct convolution with bias, ReLU and the 2x2 / stride-2 max-pool applied to the accumulators
    (`pdn_conv2d_relu_pool_fwd_f32`): only the pooled map and a hit map of one bit per position reach HBM.
    Backward: the pooled gradient is expanded through that mask -- every window position that equals the maximum
    and passes relu'(y) = [y >= 0] receives it, exactly what the reference's maximum / max grad_fns produce
    (tensor.py:808-815) -- while the data-gradient and weight-gradient kernels stage it into LDS; the
    full-resolution conv output, its relu, and both of their gradients never exist."""

    def __init__(self, x, kernel, bias=None, padding=0, stride=1):
        self.padding, self.stride = int(padding), int(stride)
        self.has_bias = bias is not None
        _Operator.__init__(self, *((x, kernel, bias) if self.has_bias else (x, kernel)))

    def forward_(self, x, kernel, bias=None):
        hp, L = _hip(), _L()
        N, C, H, W, O, k, oh, ow = self._dims(x, kernel)
        self._xd = _contig(x.data)
        self._k_shape = tuple(kernel.shape)
        self._kernel_data = kernel.data
        self._bias_data = bias.data if bias is not None else None
        self._colp = self._wp = None
        self._direct = L.query("pdn_conv2d_direct_supported", C, H, W, O, k, self.stride, self.padding)
        self._fused = L.query("pdn_conv2d_relu_pool_supported", C, H, W, O, k, self.stride, self.padding)
        out = hp.empty((N, O, oh // 2, ow // 2), np.float32)
        self._mask = hp.empty((N, O, oh * ow // 32), np.int32)              # hit map: one BIT per conv output position
        L.call("pdn_conv2d_relu_pool_fwd_f32", self._xd._ptr, _contig(kernel.data)._ptr,
               _contig(bias.data)._ptr if bias is not None else None, out._ptr, self._mask._ptr, N, C, H, W, O, k,
               self.stride, self.padding, hp.stream())
        return out

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, kernel = self.last[0], self.last[1]
        bias = self.last[2] if self.has_bias else None
        N, C, H, W, O, k, oh, ow = self._dims(x, kernel)
        gp = _contig(g)
        need_dw = kernel.requires_grad
        need_db = bias is not None and bias.requires_grad
        need_dx = x.requires_grad
        fused_w = bool(self._fused & 4) or not (need_dw or need_db)
        fused_x = bool(self._fused & 2) or not need_dx
        if not (fused_w and fused_x):
            # a direction the expanding loads do not take: materialise the expanded gradient once, plain kernels
            dy = hp.empty((N, O, oh, ow), np.float32)
            L.call("pdn_pool_mask_expand_f32", gp._ptr, self._mask._ptr, dy._ptr, N * O, oh, ow, hp.stream())
            return conv2d.backward_all(self, dy)
        grads = [None] * len(self.last)
        if need_dw or need_db:
            direct_w = need_dw and _is_leaf_f32(kernel)
            direct_b = need_db and _is_leaf_f32(bias)
            if need_dw and need_db and direct_w != direct_b:
                direct_w = direct_b = False
            dw = (kernel.grad if direct_w else hp.empty(kernel.shape, np.float32)) if need_dw else None
            db = (bias.grad if direct_b else hp.empty((O,), np.float32)) if need_db else None
            ws, wsb = hp.workspace(L.query("pdn_conv2d_bwd_weight_workspace_bytes", N, C, H, W, O, k,
                                           self.stride, self.padding))
            L.call("pdn_conv2d_relu_pool_bwd_weight_f32", self._xd._ptr, gp._ptr, self._mask._ptr,
                   dw._ptr if dw is not None else None, db._ptr if db is not None else None,
                   1 if (direct_w or direct_b) else 0, N, C, H, W, O, k, self.stride, self.padding, ws, wsb, hp.stream())
            if need_dw and not direct_w:
                grads[1] = dw
            if need_db and not direct_b:
                grads[2] = db.reshape(bias.shape)
        if need_dx:
            dx = hp.empty((N, C, H, W), np.float32)
            L.call("pdn_conv2d_relu_pool_bwd_data_f32", gp._ptr, self._mask._ptr, _contig(kernel.data)._ptr, dx._ptr,
                   N, C, H, W, O, k, self.stride, self.padding, hp.stream())
            grads[0] = dx
        return grads


class pool2d(_Operator):
    """max / avg pooling over k x k windows of the zero-padded input (nn/functional.py:284-339)."""

    def __init__(self, x, kernel_size, stride, padding=0, mode="max"):
        self.k, self.stride, self.padding = int(kernel_size), int(stride), int(padding)
        self.mode = mode
        super().__init__(x)

    def _windows(self, xd):
        p, s, k = self.padding, self.stride, self.k
        xp_ = np.pad(xd, [(0, 0), (0, 0), (p, p), (p, p)], "constant")
        N, C, H, W = xp_.shape
        oh, ow = (H - k) // s + 1, (W - k) // s + 1
        s0, s1, s2, s3 = xp_.strides
        return xp_, np.lib.stride_tricks.as_strided(xp_, (N, C, oh, ow, k, k), (s0, s1, s2 * s, s3 * s, s2, s3))

    def forward_(self, x):
        N, C, H, W = x.shape
        if self.xp is np:
            _, win = self._windows(x.data)
            return win.max((-1, -2)) if self.mode == "max" else win.mean((-1, -2))
        _require_f32(self, x)
        hp, L = _hip(), _L()
        self._x = _contig(x.data)
        oh = (H + 2 * self.padding - self.k) // self.stride + 1
        ow = (W + 2 * self.padding - self.k) // self.stride + 1
        out = hp.empty((N, C, oh, ow), np.float32)
        L.call("pdn_pool2d_fwd_f32", self._x._ptr, N, C, H, W, self.k, self.stride, self.padding,
               0 if self.mode == "max" else 1, out._ptr, hp.stream())
        return out

    def backward_all(self, g):
        x = self.last[0]
        N, C, H, W = x.shape
        if self.xp is np:
            p = self.padding
            xpad, win = self._windows(x.data)
            dxp = np.zeros(xpad.shape, dtype=g.dtype)
            s0, s1, s2, s3 = dxp.strides
            s = self.stride
            view = np.lib.stride_tricks.as_strided(dxp, win.shape, (s0, s1, s2 * s, s3 * s, s2, s3))
            if self.mode == "max":
                contrib = (win == self.data[..., None, None]) * g[..., None, None]
            else:
                contrib = np.broadcast_to(g[..., None, None] / (self.k * self.k), win.shape)
            np.add.at(view, (...,), contrib)
            return [dxp[:, :, p:p + H, p:p + W] if p else dxp]
        hp, L = _hip(), _L()
        dx = hp.empty(x.shape, np.float32)
        y, g = _contig(self.data), _contig(g)
        L.call("pdn_pool2d_bwd_f32", self._x._ptr, y._ptr, g._ptr, N, C, H, W,
               self.k, self.stride, self.padding, 0 if self.mode == "max" else 1, dx._ptr, hp.stream())
        return [dx]
