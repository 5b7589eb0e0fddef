"""Elementwise and row-wise nodes: gated_sigmoid, swiglu, silu, softmax, rope.
(One module of `pydynet_amd.core.fused`; the package docstring lists the reference chains each node replaces.)"""
from __future__ import annotations

import numpy as np

from ..tensor import Tensor, _Operator
from ._common import _hip, _L, _contig, _require_f32, _Deferred


class gated_sigmoid(_Operator):
    """y = x * sigmoid(alpha * x): CLIP's quick-GELU with alpha = 1.702 (llm/clip/model.py:92-95)."""

    def __init__(self, x, alpha=1.702):
        self.alpha = float(alpha)
        super().__init__(x)

    def forward_(self, x):
        if self.xp is np:
            return x.data / (1 + np.exp(-np.asarray(self.alpha, x.dtype) * x.data))
        _require_f32(self, x)
        hp, L = _hip(), _L()
        self._x = _contig(x.data)
        out = hp.empty(x.shape, np.float32)
        L.call("pdn_gated_sigmoid_fwd_f32", self._x._ptr, out._ptr, self.alpha, out.size, hp.stream())
        return out

    def backward_all(self, dy):
        x = self.last[0]
        if self.xp is np:
            a = np.asarray(self.alpha, x.dtype)
            s = 1 / (1 + np.exp(-a * x.data))
            return [dy * s * (1 + a * x.data * (1 - s))]
        hp, L = _hip(), _L()
        dy = _contig(dy)
        dx = hp.empty(x.shape, np.float32)
        L.call("pdn_gated_sigmoid_bwd_f32", self._x._ptr, dy._ptr, dx._ptr, self.alpha, dy.size, hp.stream())
        return [dx]


class swiglu(_Operator):
    """y = silu(gate) * up,  silu(g) = g / (1 + exp(-g))."""

    def forward_(self, gate, up):
        if self.xp is np:
            return gate.data / (1 + np.exp(-gate.data)) * up.data
        _require_f32(self, gate, up)
        hp, L = _hip(), _L()
        self._g, self._u = _contig(gate.data), _contig(up.data)
        out = hp.empty(gate.shape, np.float32)
        L.call("pdn_swiglu_fwd_f32", self._g._ptr, self._u._ptr, out._ptr, out.size, hp.stream())
        return out

    def backward_all(self, dy):
        gate, up = self.last
        if self.xp is np:
            s = 1 / (1 + np.exp(-gate.data))
            return [dy * up.data * s * (1 + gate.data * (1 - s)), dy * gate.data * s]
        hp, L = _hip(), _L()
        dy = _contig(dy)
        dg, du = hp.empty(gate.shape, np.float32), hp.empty(gate.shape, np.float32)
        L.call("pdn_swiglu_bwd_f32", self._g._ptr, self._u._ptr, dy._ptr, dg._ptr, du._ptr, dy.size, hp.stream())
        return [dg, du]


class silu(_Deferred, _Operator):
    """x * sigmoid(x) (nn/functional.py:39-40).  On a HIP device the node is created without running (`_Deferred`): the
    product `F.silu(gate) * up` of llm/llama/model.py:56-58 takes it over as ONE `swiglu` node (core/fused/chain.py:
    on_mul); any other consumer reads `.data`, which runs the kernel then."""

    defer = True               # class switch: False runs the activation at construction (no silu * up fusion)
    _mul_hook = True           # Tensor.__mul__ asks core/fused/chain.py about pending activations

    def __init__(self, x):
        if silu.defer and isinstance(x, Tensor) and x.device.is_hip and x.dtype == np.float32:
            self._init_deferred([x], x.shape, np.float32)
        else:
            super().__init__(x)

    def forward_(self, x):
        if self.xp is np:
            return x.data / (1 + np.exp(-x.data))
        _require_f32(self, x)
        hp, L = _hip(), _L()
        self._x = _contig(x.data)
        out = hp.empty(x.shape, np.float32)
        L.call("pdn_swiglu_fwd_f32", self._x._ptr, None, out._ptr, out.size, hp.stream())
        return out

    def backward_all(self, dy):
        x = self.last[0]
        if self.xp is np:
            s = 1 / (1 + np.exp(-x.data))
            return [dy * s * (1 + x.data * (1 - s))]
        hp, L = _hip(), _L()
        dy = _contig(dy)
        dx = hp.empty(x.shape, np.float32)
        L.call("pdn_swiglu_bwd_f32", self._x._ptr, None, dy._ptr, dx._ptr, None, dy.size, hp.stream())
        return [dx]


class softmax(_Operator):
    """softmax over the LAST axis (x - rowmax; the max is not differentiated, as in the reference)."""

    def forward_(self, x):
        if self.xp is np:
            e = np.exp(x.data - x.data.max(-1, keepdims=True))
            return e / e.sum(-1, keepdims=True)
        _require_f32(self, x)
        hp, L = _hip(), _L()
        xd = _contig(x.data)
        cols = x.shape[-1]
        out = hp.empty(x.shape, np.float32)
        L.call("pdn_softmax_fwd_f32", xd._ptr, out._ptr, xd.size // cols, cols, 1.0, 0, 0, hp.stream())
        return out

    def backward_all(self, dy):
        y = self.data
        if self.xp is np:
            return [(dy - (dy * y).sum(-1, keepdims=True)) * y]
        hp, L = _hip(), _L()
        dy = _contig(dy)
        cols = y.shape[-1]
        dx = hp.empty(y.shape, np.float32)
        L.call("pdn_softmax_bwd_f32", y._ptr, dy._ptr, dx._ptr, y.size // cols, cols, 1.0, hp.stream())
        return [dx]


class rope(_Operator):
    """Rotary embedding on interleaved pairs; x: (B, L, H, hd), cos/sin: (L, hd/2) (no grad)."""

    def __init__(self, x, cos, sin):
        self._cos, self._sin = cos, sin
        super().__init__(x)

    def _apply(self, a, sign):
        B, Lq, H, hd = a.shape
        cos, sin = self._cos.data, self._sin.data
        if self.xp is np:
            r, i = a[..., 0::2], a[..., 1::2]
            c, s = cos[None, :, None, :], sign * sin[None, :, None, :]
            out = np.empty(a.shape, dtype=a.dtype)
            out[..., 0::2] = r * c - i * s
            out[..., 1::2] = r * s + i * c
            return out
        _require_f32(self, self._cos, self._sin)
        if a.dtype != np.float32:
            raise TypeError(f"rope: the fused HIP kernel is float32-only, got {a.dtype}")
        hp, L = _hip(), _L()
        a, cos, sin = _contig(a), _contig(cos), _contig(sin)   # locals keep any copies alive
        out = hp.empty(a.shape, np.float32)
        L.call("pdn_rope_f32", a._ptr, cos._ptr, sin._ptr, out._ptr, B * Lq, Lq, H, hd,
               1 if sign < 0 else 0, hp.stream())
        return out

    def forward_(self, x):
        return self._apply(x.data, 1.0)

    def backward_all(self, dy):
        return [self._apply(dy, -1.0)]
