"""Normalisation nodes: rms_norm, layer_norm (leading-axis quirk of the reference), col_norm.
(One module of `pydynet_amd.core.fused`; the package docstring lists the reference chains each node replaces.)"""
from __future__ import annotations

import numpy as np

import os

from ...autograd import is_grad_enable
from ..tensor import _Operator
from ._common import _hip, _L, _contig, _require_f32, _foldable, _is_leaf_f32, _Deferred


class rms_norm(_Deferred, _Operator):
    """y = x / sqrt(mean(x^2, -1) + eps) * w   (w 1-D over the last axis).

    Round 5: on the HIP device, at the model width the row-resident projection kernels hold in registers (288) and with
    enough rows for them, the node is DEFERRED (`_Deferred`): a `qkv_attention` / `ffn_swiglu` node that consumes it
    normalises the rows inside its projection's A load (`pdn_*_norm_fwd_f32`) and hands this node its output and the rows'
    rms (`_adopt`); any other consumer reads `.data`, which runs the node's own kernel then.  Backward is unchanged."""

    folds_existing = True
    fold = os.environ.get("PDN_NO_NORM_FOLD", "0") != "1"        # (same-box A/B switch)
    fold_min_rows = 4096

    def __init__(self, x, weight, eps=1e-6):
        self.eps = float(eps)
        if (rms_norm.fold and type(self) is rms_norm and x.device.is_hip and is_grad_enable()
                and x.dtype == np.float32 and weight.dtype == np.float32 and x.ndim >= 2 and x.shape[-1] == 288
                and weight.shape == (288,) and x.size // 288 >= rms_norm.fold_min_rows):
            self._init_deferred((x, weight), x.shape, np.float32)
        else:
            super().__init__(x, weight)

    def _adopt(self, raw, rms, out):
        """A consumer's projection kernel formed this node's output: `raw` = the contiguous input rows it read."""
        self._x, self._rms = raw, rms
        self.data = out                     # (no longer pending: an ordinary node from here on)

    def forward_(self, x, w):
        if self.xp is np:
            self._rms = np.sqrt((x.data * x.data).mean(-1, keepdims=True) + np.asarray(self.eps, x.dtype))
            return x.data / self._rms * w.data
        _require_f32(self, x, w)
        hp, L = _hip(), _L()
        cols = x.shape[-1]
        self._x = _contig(x.data)
        rows = self._x.size // cols
        out = hp.empty(x.shape, np.float32)
        self._rms = hp.empty((rows,), np.float32)
        L.call("pdn_rmsnorm_fwd_f32", self._x._ptr, w.data._ptr, out._ptr, self._rms._ptr, rows, cols,
               self.eps, hp.stream())
        return out

    def backward_all(self, g):
        x, w = self.last
        if self.xp is np:
            z = x.data / self._rms
            dz = g * w.data
            dx = (dz - z * (z * dz).mean(-1, keepdims=True)) / self._rms
            return [dx if x.requires_grad else None,
                    (g * z).reshape(-1, w.shape[-1]).sum(0) if w.requires_grad else None]
        hp, L = _hip(), _L()
        cols = x.shape[-1]
        rows = self._x.size // cols
        g = _contig(g)
        dx = hp.empty(x.shape, np.float32)
        direct = w.requires_grad and _is_leaf_f32(w)
        dw = w.grad if direct else (hp.empty((cols,), np.float32) if w.requires_grad else None)
        ws, wsb = hp.workspace(L.query("pdn_rmsnorm_bwd_workspace_bytes", rows, cols))
        ex = _foldable(self, 0, x) if x.requires_grad else None
        L.call("pdn_rmsnorm_bwd_f32", self._x._ptr, w.data._ptr, self._rms._ptr, g._ptr,
               ex._ptr if ex is not None else None, dx._ptr,
               dw._ptr if dw is not None else None, 1 if direct else 0, rows, cols, ws, wsb, hp.stream())
        return [dx if x.requires_grad else None, None if direct else dw]


class layer_norm(_Operator):
    """LayerNorm over the LAST axis: (x - mean) / sqrt(var + eps) * scale + shift -- the CLIPLayerNorm of
    llm/clip/model.py:66-80 (9 generic nodes there).  The reference's own nn.LayerNorm, which reduces
    over the leading axes, is `col_norm`."""

    folds_existing = True

    def __init__(self, x, scale, shift, eps=1e-5):
        self.eps = float(eps)
        super().__init__(x, scale, shift)

    def forward_(self, x, w, b):
        if self.xp is np:
            mu = x.data.mean(-1, keepdims=True)
            self._c = x.data - mu
            self._sd = np.sqrt(np.square(self._c).mean(-1, keepdims=True) + np.asarray(self.eps, x.dtype))
            return self._c / self._sd * w.data + b.data
        _require_f32(self, x, w, b)
        hp, L = _hip(), _L()
        cols = x.shape[-1]
        self._x = _contig(x.data)
        rows = self._x.size // cols
        out = hp.empty(x.shape, np.float32)
        self._mean, self._rstd = hp.empty((rows,), np.float32), hp.empty((rows,), np.float32)
        L.call("pdn_layernorm_fwd_f32", self._x._ptr, _contig(w.data)._ptr, _contig(b.data)._ptr, out._ptr,
               self._mean._ptr, self._rstd._ptr, rows, cols, self.eps, hp.stream())
        return out

    def backward_all(self, g):
        x, w, b = self.last
        cols = x.shape[-1]
        if self.xp is np:
            xh = self._c / self._sd
            dz = g * w.data
            dx = (dz - dz.mean(-1, keepdims=True) - xh * (dz * xh).mean(-1, keepdims=True)) / self._sd
            return [dx if x.requires_grad else None,
                    (g * xh).reshape(-1, cols).sum(0).reshape(w.shape) if w.requires_grad else None,
                    g.reshape(-1, cols).sum(0).reshape(b.shape) if b.requires_grad else None]
        hp, L = _hip(), _L()
        rows = self._x.size // cols
        g = _contig(g)
        dx = hp.empty(x.shape, np.float32)
        dw_direct = w.requires_grad and _is_leaf_f32(w)
        db_direct = b.requires_grad and _is_leaf_f32(b)
        direct = dw_direct and db_direct            # one accumulate flag for both
        dw = (w.grad if direct else hp.empty((cols,), np.float32)) if w.requires_grad else None
        db = (b.grad if direct else hp.empty((cols,), np.float32)) if b.requires_grad else None
        ws, wsb = hp.workspace(L.query("pdn_layernorm_bwd_workspace_bytes", rows, cols))
        ex = _foldable(self, 0, x) if x.requires_grad else None
        L.call("pdn_layernorm_bwd_f32", self._x._ptr, _contig(w.data)._ptr, self._mean._ptr, self._rstd._ptr, g._ptr,
               ex._ptr if ex is not None else None, dx._ptr, dw.reshape(-1)._ptr if dw is not None else None,
               db.reshape(-1)._ptr if db is not None else None, 1 if direct else 0, rows, cols, ws, wsb, hp.stream())
        return [dx if x.requires_grad else None,
                None if (direct or dw is None) else dw.reshape(w.shape),
                None if (direct or db is None) else db.reshape(b.shape)]


class col_norm(_Operator):
    """Reference LayerNorm / BatchNorm1d in training mode (nn/modules/norm.py:60-74, 203-218):
    per-column statistics of x viewed as (rows, cols) -- the reference's LayerNorm reduces over the
    LEADING axes -- then `(x - mean) / sqrt(var + eps) * scale + shift`, with the running statistics
    updated in the same launch sequence.  HIP device only; the NumPy device composes generic ops."""

    def __init__(self, x, scale, shift, running_mean, running_var, eps, momentum, cols):
        self._rm, self._rv, self.eps, self.momentum, self.cols = running_mean, running_var, float(eps), float(momentum), int(cols)
        super().__init__(x, scale, shift)

    def forward_(self, x, scale, shift):
        if self.xp is np:
            raise NotImplementedError("col_norm is the HIP fused path")
        _require_f32(self, x, scale, shift, self._rm, self._rv)
        hp, L = _hip(), _L()
        cols = self.cols
        xd = _contig(x.data)
        rows = xd.size // cols
        y = hp.empty(x.shape, np.float32)
        mean, rstd = hp.empty((cols,), np.float32), hp.empty((cols,), np.float32)
        ws, wsb = hp.workspace(L.query("pdn_colnorm_workspace_bytes", rows, cols))
        rm, rv = self._rm.data, self._rv.data
        L.call("pdn_colnorm_fwd_f32", xd._ptr, _contig(scale.data)._ptr, _contig(shift.data)._ptr, y._ptr,
               mean._ptr, rstd._ptr, rm._ptr, rv._ptr, self.momentum, self.eps, rows, cols, ws, wsb, hp.stream())
        self._saved = (xd, mean, rstd, rows)
        return y

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, scale, shift = self.last
        xd, mean, rstd, rows = self._saved
        cols = self.cols
        g = _contig(g)
        dx = hp.empty(x.shape, np.float32) if x.requires_grad else None
        grads = [dx, None, None]
        direct_w = scale.requires_grad and _is_leaf_f32(scale)
        direct_b = shift.requires_grad and _is_leaf_f32(shift)
        acc = direct_w or direct_b
        # leaf buffers are accumulated into directly; otherwise fresh arrays are returned
        dw = scale.grad if direct_w else (hp.zeros((cols,), np.float32) if scale.requires_grad else None)
        db = shift.grad if direct_b else (hp.zeros((cols,), np.float32) if shift.requires_grad else None)
        ws, wsb = hp.workspace(L.query("pdn_colnorm_workspace_bytes", rows, cols))
        L.call("pdn_colnorm_bwd_f32", xd._ptr, _contig(scale.data)._ptr, mean._ptr, rstd._ptr, g._ptr,
               dx._ptr if dx is not None else None, dw.reshape(-1)._ptr if dw is not None else None,
               db.reshape(-1)._ptr if db is not None else None, 1 if acc else 0, rows, cols, ws, wsb, hp.stream())
        if scale.requires_grad and not direct_w:
            grads[1] = dw.reshape(scale.shape)
        if shift.requires_grad and not direct_b:
            grads[2] = db.reshape(shift.shape)
        return grads
