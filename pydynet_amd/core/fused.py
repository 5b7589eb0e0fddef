"""Fused differentiable nodes of the training hot path.

Each class is an ordinary tape node (same protocol as the reference's operators) that stands
for a chain of generic nodes in the reference and runs as ONE forward and ONE backward HIP
kernel (or GEMM) on a GPU device; on "cpu" the same node evaluates the equivalent NumPy
expression.  Reference chains replaced:

    linear          nn/functional.py:7-11           (matmul + broadcast add; dW summed by the engine)
    rms_norm        nn/modules/norm.py:245-248      (6 nodes)
    silu / swiglu   nn/functional.py:39-40, llm/llama/model.py:56-58
    relu            nn/functional.py:31-32          (maximum(0., x); gradient 1 at x == 0)
    softmax         nn/functional.py:43-49          (last axis)
    rope            llm/llama/model.py:23-44        (26 nodes)
    attention       llm/llama/model.py:112-121      (transpose, matmul, /sqrt(hd), +mask, softmax, matmul)
    embedding       nn/functional.py:14-20 + tensor.py:937-940 (scatter-ASSIGN gradient)
    cross_entropy   nn/functional.py:364-381        (7 nodes, integer targets)
"""
from __future__ import annotations

import contextlib
import math
import os

import numpy as np

from ..autograd import is_grad_enable
from .tensor import Graph, Tensor, _Operator, _as_operand


def _hip():
    from .. import hipnp
    return hipnp


def _L():
    from .. import _lib
    return _lib.lib()


def _contig(a):
    return a if a.is_contiguous() else a.copy()


def hip_f32(*tensors):
    """True when these operands may take a fused HIP node: the kernels behind them are float32
    only, so on a HIP device every (non-None) operand must be float32.  On "cpu" the nodes are NumPy
    expressions and carry any floating dtype."""
    dev = next(t for t in tensors if t is not None).device
    if not dev.is_hip:
        return True
    return all(t is None or (t.dtype == np.float32 and t.device == dev) for t in tensors)


def _require_f32(node, *tensors):
    """Fused HIP kernels reinterpret raw buffers as float32: refuse anything else loudly."""
    for t in tensors:
        if t is not None and t.dtype != np.float32:
            raise TypeError(f"{type(node).__name__}: the fused HIP kernel is float32-only, got {t.dtype} "
                            "(use the generic operators, or cast with .astype(np.float32))")


def _foldable(node, idx, t):
    """The gradient input `idx` already holds (handed over by the engine as `node._existing`) if
    this node can add it inside its own kernel; marks the input as folded."""
    ex = getattr(node, "_existing", None)
    ex = ex[idx] if ex is not None else None
    if ex is None or isinstance(ex, np.ndarray) or ex.dtype != np.float32 \
            or ex.shape != tuple(t.shape) or not ex.is_contiguous():
        return None
    node._folded.add(idx)
    return ex


two_stream = {"enabled": os.environ.get("PDN_TWO_STREAM", "0") == "1"}


def _beside(hp, fin, fout, has_dx):
    """A `hipnp.side_stream` for the weight-gradient product of a projection, to run beside the
    input-gradient product.  Opt-in (PDN_TWO_STREAM=1): isolated 768-wide pairs gain 7-8 % (one GEMM's store
    tail under the other's main loop, `tools/two_stream_probe.py`), but a whole training step measured
    +-0.5 % (same-box A/B at per-GPU batch 64 / 128 / 256), i.e. nothing."""
    if not (two_stream["enabled"] and has_dx and 512 <= max(fin, fout) <= 4096) or hp.capturing() is not None:
        return None
    return hp.side_stream()


def _is_leaf_f32(t):
    return (t.requires_grad and not t.last and t.grad is not None and t.grad.dtype == np.float32
            and (isinstance(t.grad, np.ndarray) or t.grad.is_contiguous()))


# ---------------------------------------------------------------------------------------
class _Deferred:
    """Mixin for a node whose array is produced at FIRST USE instead of at construction.

    The reference composes `max_pool2d(relu(conv2d(x)), 2, 2)` from three tape nodes (mnist.py:92-95), each a full pass
    over HBM.  A conv2d node whose shape the fused kernel takes is created without running anything; `relu` of such a
    node is deferred too; `max_pool2d(., 2, 2)` of that then launches ONE kernel (conv + bias + relu + pool in the
    epilogue, `conv2d_relu_pool`) and the two intermediate nodes are simply dropped.  Any other consumer reads `.data`,
    which runs the node's own kernel then -- from that moment it is an ordinary node.  Metadata (`shape`, `dtype`,
    ...) is answered without materialising.  (As with any lazy value: inputs modified in place between construction
    and first use are seen in their modified state.)"""

    _pending = None

    def _init_deferred(self, inputs, shape, dtype):
        self._pending, self._shape, self._dtype = tuple(inputs), tuple(int(v) for v in shape), np.dtype(dtype)
        self.__dict__["_data"] = None
        self.device = inputs[0].device
        self.copy = None
        self.grad = None
        self.requires_grad = bool(is_grad_enable() and any(t.requires_grad for t in inputs))
        self.last = list(inputs) if self.requires_grad else []
        if self.requires_grad:
            Graph._add_node(self)

    @property
    def data(self):
        d = self.__dict__.get("_data")
        if d is None and self._pending is not None:
            inputs, self._pending = self._pending, None
            with self.device:
                d = self.forward_(*inputs)
            self.__dict__["_data"] = d
        return d

    @data.setter
    def data(self, value):
        self.__dict__["_data"] = value
        self._pending = None

    @property
    def shape(self): return self._shape if self._pending is not None else self.data.shape
    @property
    def dtype(self): return self._dtype if self._pending is not None else self.data.dtype
    @property
    def ndim(self): return len(self.shape)
    @property
    def size(self): return int(np.prod(self.shape, dtype=np.int64))


class linear(_Operator):
    """y = x @ W (+ b) (+ residual) over the last axis of x; W is (in, out).

    `residual` (shape of y) folds the `z = x + sublayer(x)` add of a transformer block into the
    GEMM epilogue; its gradient is the upstream gradient itself."""

    folds_existing = True      # backward adds the gradient x already holds inside the dX GEMM

    def __init__(self, x, weight, bias=None, residual=None):
        self.has_bias, self.has_res = bias is not None, residual is not None
        ins = [x, weight] + ([bias] if self.has_bias else []) + ([residual] if self.has_res else [])
        super().__init__(*ins)

    def _split(self, ins):
        b = ins[2] if self.has_bias else None
        r = ins[2 + self.has_bias] if self.has_res else None
        return ins[0], ins[1], b, r

    def forward_(self, *ins):
        x, w, b, r = self._split(ins)
        if self.xp is np:
            y = x.data @ w.data
            if b is not None:
                y = y + b.data
            return y + r.data if r is not None else y
        _require_f32(self, x, w, b, r)
        hp = _hip()
        fin, fout = w.shape
        x2 = x.data.reshape(-1, fin)
        out = hp.empty(x.shape[:-1] + (fout,), np.float32)
        res = _contig(r.data).reshape(-1, fout) if r is not None else None
        hp.gemm(x2, w.data, out.reshape(-1, fout), bias=b.data.reshape(-1) if b is not None else None,
                residual=res)
        return out

    def _dx(self, hp, g2, x, w, fin):
        dx = hp.empty(x.shape, np.float32)
        ex = _foldable(self, 0, x)
        hp.gemm(g2, w.data.T, dx.reshape(-1, fin),                         # NT
                residual=ex.reshape(-1, fin) if ex is not None else None)
        return dx

    def backward_all(self, g):
        x, w, b, r = self._split(self.last)
        fin, fout = w.shape
        grads = [None] * len(self.last)
        if self.has_res and r.requires_grad:
            grads[2 + self.has_bias] = g
        if self.xp is np:
            g2, x2 = g.reshape(-1, fout), x.data.reshape(-1, fin)
            if x.requires_grad:
                grads[0] = (g2 @ w.data.T).reshape(x.shape)
            if w.requires_grad:
                grads[1] = x2.T @ g2
            if b is not None and b.requires_grad:
                grads[2] = g2.sum(0).reshape(b.shape)
            return grads
        hp = _hip()
        g2 = _contig(g).reshape(-1, fout)
        x2 = x.data.reshape(-1, fin)
        side = _beside(hp, fin, fout, x.requires_grad and w.requires_grad)
        if x.requires_grad and side is None:
            grads[0] = self._dx(hp, g2, x, w, fin)
        need_db = b is not None and b.requires_grad
        aux = getattr(g, "_aux", None)
        if need_db and aux is not None and aux[0] == "colsum" and aux[1].size == b.size:
            # the producer of g (fused cross entropy) already summed its columns
            if _is_leaf_f32(b):
                b.grad += aux[1].reshape(b.grad.shape)
            else:
                grads[2] = aux[1].reshape(b.shape)
            need_db = False
        # bias gradient = column sums of g: formed inside the dW GEMM (both read g once) when the
        # operands have the aligned x^T @ g layout and the leaf buffers can be accumulated into
        # (fusing costs ~25 % of the dW GEMM, a separate pass over g one read of it: the fusion only
        # pays for short contractions, fin < ~8 * MFMA rate / HBM rate ~ 192)
        fuse_db = (need_db and w.requires_grad and _is_leaf_f32(b) and x2.is_contiguous() and fin < 192
                   and fin % 4 == 0 and fout % 4 == 0 and x2.shape[0] % 4 == 0)
        if w.requires_grad:
            cs = b.grad.reshape(-1) if fuse_db else None
            dw = None if _is_leaf_f32(w) else hp.empty((fin, fout), np.float32)
            with side or contextlib.nullcontext():
                if dw is None:
                    hp.gemm(x2.T, g2, w.grad, beta=1.0, b_colsum=cs, colsum_accumulate=True)   # TN, += into the leaf
                else:
                    hp.gemm(x2.T, g2, dw, b_colsum=cs, colsum_accumulate=True)
                    grads[1] = dw
        if side is not None:
            grads[0] = self._dx(hp, g2, x, w, fin)
            side.join()
        if need_db and not fuse_db:
            grads[2] = g2.sum(0).reshape(b.shape)
        return grads


class rms_norm(_Operator):
    """y = x / sqrt(mean(x^2, -1) + eps) * w   (w 1-D over the last axis)."""

    folds_existing = True

    def __init__(self, x, weight, eps=1e-6):
        self.eps = float(eps)
        super().__init__(x, weight)

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


class silu(_Operator):
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


def _attn_kernel(B, H, hd, Lq, Lk, start_pos, has_mask, layouts):
    """'resident' (K/V of a head held in LDS 256 rows at a time: hd 48 / 64, L <= 1024 -- the benchmark shape is
    one chunk), 'stream' (general kernels) or None (GEMM + softmax composition)."""
    if not attention.use_flash or any(l is None for l in layouts):
        return None
    ql, kl, vl = layouts
    dense = (H * hd, Lq * H * hd if B > 1 else 0)
    if (Lq == Lk and start_pos == 0 and not has_mask and attention.use_resident and ql == kl == vl == dense
            and _L().query("pdn_attention_supported", Lq, hd)):
        return "resident"
    if _L().query("pdn_attention_stream_supported", hd) and kl == vl:
        return "stream"
    return None


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
        self._kind = _attn_kernel(B, H, hd, Lq, Lk, self.start_pos, self._mask is not None, layouts)
        if self._kind == "resident":
            # scores stay in registers: one kernel, nothing of size L x L in HBM; lse kept for backward
            out = hp.empty((B, Lq, H, hd), np.float32)
            self._lse = hp.empty((B, H, Lq), np.float32)
            L.call("pdn_attention_fwd_f32", q.data._ptr, k.data._ptr, v.data._ptr, out._ptr, self._lse._ptr,
                   B, H, Lq, hd, H * hd, Lq * H * hd, H * hd, Lq * H * hd, causal, None, None, hp.stream())
            return out
        if self._kind == "stream":
            out = hp.empty((B, Lq, H, hd), np.float32)
            self._lse = hp.empty((B, H, Lq), np.float32)
            mp, sb, sh, sq, sk, self._mask_dev = _attn_mask_args(
                hp.asarray(self._mask) if self._mask is not None else None, B, H, Lq, Lk)
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
            do = _contig(do)
            dq, dk, dv = (hp.empty(q.shape, np.float32) for _ in range(3))
            ws, wsb = hp.workspace(L.query("pdn_attention_bwd_workspace_bytes", B, H, Lq))
            L.call("pdn_attention_bwd_f32", q.data._ptr, k.data._ptr, v.data._ptr, self.data._ptr, do._ptr,
                   self._lse._ptr, dq._ptr, dk._ptr, dv._ptr, B, H, Lq, hd, H * hd, Lq * H * hd, H * hd, Lq * H * hd,
                   causal, None, None, ws, wsb, hp.stream())
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


class embedding(_Operator):
    """out = W[ids]; gradient = scatter-ASSIGN of the last occurrence of each id (reference
    semantics, tensor.py:937-940); `accumulate=True` opts into torch-style scatter-add."""

    accumulate = False

    def __init__(self, ids, weight):
        if isinstance(ids, Tensor):
            ids = ids.data
        self._ids = ids
        super().__init__(weight)

    def forward_(self, w):
        if self.xp is np:
            return w.data[self._ids]
        hp = _hip()
        if isinstance(self._ids, np.ndarray) or not hasattr(self._ids, "_ptr"):
            self._ids = hp.from_numpy(np.asarray(self._ids).astype(np.int64))
        return w.data[self._ids]

    def backward_all(self, g):
        w = self.last[0]
        # data parallel (distributed.DataParallel): per-row owner rank, so that the summed gradient keeps
        # the scatter-ASSIGN semantics of the concatenated batch; irrelevant for scatter-add
        owner = tag = None
        if getattr(w, "_dp_owner", None) is not None and not self.accumulate:
            owner, tag = w._dp_owner(self._ids, w.shape[0])
        if self.xp is np:
            full = np.zeros(w.shape, dtype=w.dtype)
            if self.accumulate:
                np.add.at(full, self._ids, g)
            elif owner is not None:
                ids = np.asarray(self._ids).reshape(-1)
                keep = owner[ids] == tag
                full[ids[keep]] = g.reshape(ids.size, -1)[keep]
            else:
                full[self._ids] = g
            return [full]
        hp, L = _hip(), _L()
        V, D = w.shape
        g = _contig(g)
        ids = _contig(self._ids)
        ws, wsb = hp.workspace(V * 4)
        optr, otag = (owner._ptr, tag) if owner is not None else (None, 0.0)
        if _is_leaf_f32(w):
            L.call("pdn_embedding_scatter_f32", g._ptr, ids._ptr, ids.size, w.grad._ptr, V, D,
                   2 if self.accumulate else 1, optr, otag, ws, wsb, hp.stream())
            return [None]
        full = hp.zeros(w.shape, np.float32)
        L.call("pdn_embedding_scatter_f32", g._ptr, ids._ptr, ids.size, full._ptr, V, D,
               2 if self.accumulate else 0, optr, otag, ws, wsb, hp.stream())
        return [full]


class cross_entropy(_Operator):
    """mean / sum over rows of  logsumexp(x_n) - x_n[t_n]   (integer targets)."""

    def __init__(self, logits, targets, reduction="mean"):
        if reduction not in ("mean", "sum"):
            raise ValueError("reduction must be mean or sum.")
        self.reduction = reduction
        self._t = targets.data if isinstance(targets, Tensor) else targets
        super().__init__(logits)

    def forward_(self, x):
        n, V = x.shape
        if self.xp is np:
            t = np.asarray(self._t)
            m = x.data.max(-1, keepdims=True)
            self._lse = np.log(np.exp(x.data - m).sum(-1, keepdims=True)) + m
            rows = self._lse[:, 0] - x.data[np.arange(n), t]
            return rows.mean() if self.reduction == "mean" else rows.sum()
        _require_f32(self, x)
        hp, L = _hip(), _L()
        if not hasattr(self._t, "_ptr"):
            self._t = hp.from_numpy(np.asarray(self._t).astype(np.int64))
        self._x = _contig(x.data)
        self._t = _contig(self._t)
        loss_row = hp.empty((n,), np.float32)
        self._lse = hp.empty((n,), np.float32)
        out = hp.empty((1,), np.float32)
        mean = 1 if self.reduction == "mean" else 0
        self._dx = None
        from ..autograd import is_grad_enable
        if x.requires_grad and is_grad_enable():
            # the gradient w.r.t. the logits needs nothing computed later: write it now, while each
            # row is still in L2 (one pass over HBM for forward + backward)
            self._dx = hp.empty((n, V), np.float32)
            # column sums of dlogits (= the bias gradient of the Linear that made the logits) come
            # for free while the rows stream through; handed to that Linear via the array's `_aux`
            wsb = L.query("pdn_cross_entropy_colsum_workspace_bytes", n, V)
            cs = hp.empty((V,), np.float32) if wsb else None
            ws, wsb = hp.workspace(wsb) if wsb else (None, 0)
            L.call("pdn_cross_entropy_fwd_bwd_f32", self._x._ptr, self._t._ptr, n, V, mean,
                   1.0 / n if mean else 1.0, loss_row._ptr, self._lse._ptr, out._ptr, self._dx._ptr,
                   cs._ptr if cs is not None else None, ws, wsb, hp.err_flag_ptr(), hp.stream())
            self._dx._aux = ("colsum", cs) if cs is not None else None
        else:
            L.call("pdn_cross_entropy_fwd_f32", self._x._ptr, self._t._ptr, n, V, mean, loss_row._ptr,
                   self._lse._ptr, out._ptr, hp.err_flag_ptr(), hp.stream())
        return out.reshape(())

    def backward_all(self, g):
        x = self.last[0]
        n, V = x.shape
        scale = 1.0 / n if self.reduction == "mean" else 1.0
        if self.xp is np:
            sm = np.exp(x.data - self._lse)
            sm[np.arange(n), np.asarray(self._t)] -= 1
            return [sm * (g * np.asarray(scale, x.dtype))]
        hp, L = _hip(), _L()
        g = _contig(g)
        if self._dx is not None:
            dx, self._dx = self._dx, None       # written in forward; apply the upstream scalar (1 -> no-op)
            L.call("pdn_scale_by_device_scalar_f32", dx._ptr, dx.size, g._ptr, hp.stream())
            aux = getattr(dx, "_aux", None)
            if aux is not None:
                L.call("pdn_scale_by_device_scalar_f32", aux[1]._ptr, aux[1].size, g._ptr, hp.stream())
            return [dx]
        dx = hp.empty((n, V), np.float32)
        L.call("pdn_cross_entropy_bwd_f32", self._x._ptr, self._t._ptr, self._lse._ptr, g._ptr,
               scale, dx._ptr, n, V, hp.stream())
        return [dx]


class linear_cross_entropy(_Operator):
    """loss = cross_entropy(x @ W + b, targets) as ONE tape node (llm/llama/model.py:179 feeding
    nn/functional.py:364-381): the (rows, V) gradient of the logits never exists in memory.  Forward: the
    logits GEMM and one read-only pass for the row statistics (log-sum-exp, loss).  Backward: the two products
    `dlogits @ W^T` and `x^T @ dlogits` (and the column sums for the bias) form
    dlogits = (softmax - onehot) * scale from the saved logits as they consume it (`pdn_linear_ce_backward_f32`).
    Versus linear + cross_entropy nodes: one (rows x V) write and none of its re-reads less."""

    folds_existing = True
    enabled = True
    min_rows = int(os.environ.get("PDN_LINCE_MIN_ROWS", "16384"))
    lse_epilogue = os.environ.get("PDN_NO_LSE_EPILOGUE", "0") != "1"      # row statistics in the projection's store (A/B switch)
    # statistics split over the projection (row maxima) and the input-gradient product (sum of exponentials), the latter
    # run in the forward pass of a training step (A/B switch)
    deferred_norm = os.environ.get("PDN_NO_CE_DEFERRED", "0") != "1"

    @staticmethod
    def applicable(x, w, b, targets, reduction="mean"):
        if not (linear_cross_entropy.enabled and x.device.is_hip and x.dtype == np.float32 and w.dtype == np.float32
                and (b is None or b.dtype == np.float32) and reduction in ("mean", "sum") and w.ndim == 2):
            return False
        rows = 1
        for d in x.shape[:-1]:
            rows *= d
        t = targets.data if isinstance(targets, Tensor) else targets
        # (below 28672 tokens the row workgroups of the output-resident input-gradient kernel no longer fill the chip:
        #  it then cuts K = vocabulary into ranges over the grid, pdn_gemm_outres_plan; below `min_rows` the separate
        #  nodes on the tiled kernels are left in place)
        return (x.shape[-1] == w.shape[0] and getattr(t, "ndim", 0) == 1 and t.shape[0] == rows
                and rows >= linear_cross_entropy.min_rows
                and bool(_L().query("pdn_linear_ce_supported", rows, w.shape[1], w.shape[0])))

    def __init__(self, x, weight, bias, targets, reduction="mean"):
        self.reduction = reduction
        self.has_bias = bias is not None
        self._t = targets.data if isinstance(targets, Tensor) else targets
        super().__init__(*([x, weight] + ([bias] if self.has_bias else [])))

    def forward_(self, x, w, b=None):
        _require_f32(self, x, w, b)
        hp, L = _hip(), _L()
        fin, V = w.shape
        x2 = _contig(x.data).reshape(-1, fin)
        n = x2.shape[0]
        if not hasattr(self._t, "_ptr"):
            self._t = hp.from_numpy(np.asarray(self._t).astype(np.int64))
        self._t = _contig(self._t)
        logits = hp.empty((n, V), np.float32)
        loss_row, lse, out = hp.empty((n,), np.float32), hp.empty((n,), np.float32), hp.empty((1,), np.float32)
        wd = w.data
        in_gemm = bool(wd.is_contiguous() and x2._strides[1] == 1 and L.query("pdn_linear_lse_supported", n, V, fin))
        self.deferred = bool(linear_cross_entropy.deferred_norm and wd.is_contiguous() and x2._strides[1] == 1
                             and is_grad_enable() and x.requires_grad
                             and L.query("pdn_linear_rowmax_supported", n, V, fin)
                             and L.query("pdn_linear_ce_dx_deferred_supported", n, V, fin))
        self.stats_in_gemm = bool(linear_cross_entropy.lse_epilogue and in_gemm and not self.deferred)
        self._dxu = None
        bp = b.data._ptr if b is not None else None
        mean = 1 if self.reduction == "mean" else 0
        if self.deferred:
            # the projection leaves the row maxima; the input-gradient product -- it needs exp(logit - max) anyway, and not
            # the upstream gradient, a scalar applied in backward -- sums the exponentials as it multiplies: it runs HERE,
            # the loss follows from its log-sum-exp with one gather per row, and no pass over the logits exists
            # (few rows: both products cut the vocabulary into ranges over the grid -- `parts` vectors of maxima, a
            #  workspace of unnormalised rows and row sums)
            parts = L.query("pdn_linear_rowmax_parts", n, V, fin)
            rowmax = hp.empty((parts * n,), np.float32)
            L.call("pdn_linear_rowmax_fwd_f32", x2._ptr, wd._ptr, bp, logits._ptr, rowmax._ptr, n, V, fin, x2._strides[0],
                   V, V, hp.stream())
            self._dxu = hp.empty((n, fin), np.float32)
            ws, wsb = hp.workspace(L.query("pdn_linear_ce_dx_deferred_workspace_bytes", n, V, fin))
            L.call("pdn_linear_ce_dx_deferred_f32", logits._ptr, rowmax._ptr, parts, self._t._ptr,
                   1.0 / n if mean else 1.0, wd._ptr, self._dxu._ptr, lse._ptr, n, V, fin, ws, wsb, hp.stream())
            L.call("pdn_cross_entropy_from_lse_f32", logits._ptr, V, lse._ptr, self._t._ptr, n, V, mean, loss_row._ptr,
                   out._ptr, hp.err_flag_ptr(), hp.stream())
        elif self.stats_in_gemm:
            # the projection leaves the rows' log-sum-exp itself (transposed accumulators: a lane owns a token): no
            # pass over the logits for the statistics, the loss is one gather per row
            L.call("pdn_linear_lse_fwd_f32", x2._ptr, wd._ptr, bp, logits._ptr, lse._ptr, n, V, fin, x2._strides[0], V, V,
                   hp.stream())
            L.call("pdn_cross_entropy_from_lse_f32", logits._ptr, V, lse._ptr, self._t._ptr, n, V, mean, loss_row._ptr,
                   out._ptr, hp.err_flag_ptr(), hp.stream())
        else:
            hp.gemm(x2, wd, logits, bias=b.data.reshape(-1) if b is not None else None)
            L.call("pdn_cross_entropy_fwd_f32", logits._ptr, self._t._ptr, n, V, mean, loss_row._ptr, lse._ptr, out._ptr,
                   hp.err_flag_ptr(), hp.stream())
        self._saved = (x2, logits, lse)
        return out.reshape(())

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, w = self.last[0], self.last[1]
        b = self.last[2] if self.has_bias else None
        fin, V = w.shape
        x2, logits, lse = self._saved
        self._saved = None
        n = x2.shape[0]
        g = _contig(g)
        grads = [None] * len(self.last)
        dx = ex = None
        dxu, self._dxu = self._dxu, None
        if x.requires_grad and dxu is not None:
            dxu *= g.reshape(())                           # formed in the forward pass, up to the upstream scalar
            grads[0] = dxu.reshape(x.shape)
        elif x.requires_grad:
            dx = hp.empty(x.shape, np.float32)
            ex = _foldable(self, 0, x)
            grads[0] = dx
        dw, dw_beta = None, 0.0
        if w.requires_grad:
            if _is_leaf_f32(w):
                dw, dw_beta = w.grad, 1.0
            else:
                dw = grads[1] = hp.empty((fin, V), np.float32)
        db, db_beta = None, 0.0
        if b is not None and b.requires_grad:
            if _is_leaf_f32(b):
                db, db_beta = b.grad.reshape(-1), 1.0
            else:
                db = hp.empty((V,), np.float32)
                grads[2] = db.reshape(b.shape)
        ws, wsb = hp.workspace(L.query("pdn_linear_ce_workspace_bytes", n, V, fin)) if (dw is not None or db is not None) else (None, 0)
        L.call("pdn_linear_ce_backward_f32", x2._ptr, x2._strides[0], logits._ptr, lse._ptr, self._t._ptr,
               1.0 / n if self.reduction == "mean" else 1.0, g._ptr, w.data._ptr,
               dx._ptr if dx is not None else None, ex._ptr if ex is not None else None,
               dw._ptr if dw is not None else None, dw_beta, db._ptr if db is not None else None, db_beta,
               n, V, fin, ws, wsb, hp.stream())
        return grads


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


def _cell_grads(node, hp, x, h, wx, wh, bias, xd, hd, dlin, grads, ix=0, ih=1, iwx=2, iwh=3, ib=4):
    """Gradients every recurrent cell shares once d(pre-activation) is known: dx = dlin Wx^T,
    dh = dlin Wh^T, dWx += x^T dlin, dWh += h^T dlin, db = column sums of dlin."""
    if x.requires_grad:
        dx = hp.empty(x.shape, np.float32)
        hp.gemm(dlin, wx.data.T, dx)
        grads[ix] = dx
    if h.requires_grad:
        dh = hp.empty(h.shape, np.float32)
        hp.gemm(dlin, wh.data.T, dh)
        grads[ih] = dh if grads[ih] is None else grads[ih] + dh
    for idx, a, w in ((iwx, xd, wx), (iwh, hd, wh)):
        if not w.requires_grad:
            continue
        if _is_leaf_f32(w):
            hp.gemm(a.T, dlin, w.grad, beta=1.0)
        else:
            dw = hp.empty(w.shape, np.float32)
            hp.gemm(a.T, dlin, dw)
            grads[idx] = dw
    if bias is not None and bias.requires_grad:
        grads[ib] = dlin.sum(0).reshape(bias.shape)


class rnn_cell(_Operator):
    """One Elman step (nn/modules/rnn.py:35-47) as a single tape node on the HIP device:
    h' = act(x Wx + h Wh + b): 2 GEMMs (the second accumulating, bias in the first's epilogue) + one
    pointwise kernel forward; one pointwise kernel + 4 GEMMs backward.  act: "tanh" | "relu"."""

    def __init__(self, x, h, wx, wh, bias=None, nonlinearity="tanh"):
        self.act = {"tanh": 0, "relu": 1}[nonlinearity]
        self.has_bias = bias is not None
        super().__init__(*((x, h, wx, wh) + ((bias,) if self.has_bias else ())))

    def forward_(self, x, h, wx, wh, bias=None):
        if self.xp is np:
            raise NotImplementedError("rnn_cell is the HIP fused path; the NumPy device composes generic ops")
        _require_f32(self, x, h, wx, wh, bias)
        hp, L = _hip(), _L()
        B, H = h.shape
        xd, hd = _contig(x.data), _contig(h.data)
        lin = hp.empty((B, H), np.float32)
        hp.gemm(xd, wx.data, lin, bias=bias.data.reshape(-1) if bias is not None else None)
        hp.gemm(hd, wh.data, lin, beta=1.0)
        y = hp.empty((B, H), np.float32)
        L.call("pdn_rnn_cell_fwd_f32", lin._ptr, y._ptr, y.size, self.act, hp.stream())
        self._saved = (xd, hd, lin)
        return y

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, h, wx, wh = self.last[:4]
        bias = self.last[4] if self.has_bias else None
        xd, hd, lin = self._saved
        g = _contig(g)
        dlin = hp.empty(lin.shape, np.float32)
        L.call("pdn_rnn_cell_bwd_f32", lin._ptr, self.data._ptr, g._ptr, dlin._ptr, dlin.size, self.act, hp.stream())
        grads = [None] * len(self.last)
        _cell_grads(self, hp, x, h, wx, wh, bias, xd, hd, dlin, grads)
        return grads


class lstm_cell(_Operator):
    """One LSTM step (nn/modules/rnn.py:244-262) as a single tape node on the HIP device.  The node's value
    is the packed pair (B, 2H) = [h' | c'] (the module hands out the two halves as views); 2 GEMMs + one
    pointwise kernel forward (12 generic nodes in the reference), one pointwise kernel + 4 GEMMs backward."""

    def __init__(self, x, h, c, wx, wh, bias=None):
        self.has_bias = bias is not None
        super().__init__(*((x, h, c, wx, wh) + ((bias,) if self.has_bias else ())))

    def forward_(self, x, h, c, wx, wh, bias=None):
        if self.xp is np:
            raise NotImplementedError("lstm_cell is the HIP fused path; the NumPy device composes generic ops")
        _require_f32(self, x, h, c, wx, wh, bias)
        hp, L = _hip(), _L()
        B, H = h.shape
        xd, hd, cd = _contig(x.data), _contig(h.data), _contig(c.data)
        lin = hp.empty((B, 4 * H), np.float32)
        hp.gemm(xd, wx.data, lin, bias=bias.data.reshape(-1) if bias is not None else None)
        hp.gemm(hd, wh.data, lin, beta=1.0)
        gates, tc, hc = hp.empty((B, 4 * H), np.float32), hp.empty((B, H), np.float32), hp.empty((B, 2 * H), np.float32)
        L.call("pdn_lstm_cell_fwd_f32", lin._ptr, cd._ptr, gates._ptr, tc._ptr, hc._ptr, B, H, hp.stream())
        self._saved = (xd, hd, cd, gates, tc)
        return hc

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, h, c, wx, wh = self.last[:5]
        bias = self.last[5] if self.has_bias else None
        xd, hd, cd, gates, tc = self._saved
        B, H = hd.shape
        g = _contig(g)
        dlin, dc = hp.empty((B, 4 * H), np.float32), hp.empty((B, H), np.float32)
        L.call("pdn_lstm_cell_bwd_f32", g._ptr, gates._ptr, tc._ptr, cd._ptr, dlin._ptr, dc._ptr, B, H, hp.stream())
        grads = [None] * len(self.last)
        if c.requires_grad:
            grads[2] = dc
        _cell_grads(self, hp, x, h, wx, wh, bias, xd, hd, dlin, grads, ix=0, ih=1, iwx=3, iwh=4, ib=5)
        return grads


class gru_cell(_Operator):
    """One GRU step (nn/modules/rnn.py:537-544) as a single tape node on the HIP device:
        [z, r] = sigmoid(x Wx1 + h Wh1 + b1);  n = tanh(x Wx2 + (r*h) Wh2 + b2);  h' = (1-z) h + z n
    4 GEMMs + 2 gate kernels forward, 8 GEMMs + 2 gate kernels backward (the generic composition is
    ~20 nodes / ~45 launches per step).  Inputs: x (B, in), h (B, H), Wx1, Wh1, Wx2, Wh2[, b1, b2]."""

    def __init__(self, x, h, wx1, wh1, wx2, wh2, b1=None, b2=None):
        self.has_bias = b1 is not None
        super().__init__(*((x, h, wx1, wh1, wx2, wh2) + ((b1, b2) if self.has_bias else ())))

    def forward_(self, x, h, wx1, wh1, wx2, wh2, b1=None, b2=None):
        if self.xp is np:
            raise NotImplementedError("gru_cell is the HIP fused path; the NumPy device composes generic ops")
        _require_f32(self, x, h, wx1, wh1, wx2, wh2, b1, b2)
        hp, L = _hip(), _L()
        B, H = h.shape
        xd, hd = _contig(x.data), _contig(h.data)
        g1 = hp.empty((B, 2 * H), np.float32)
        hp.gemm(xd, wx1.data, g1, bias=b1.data.reshape(-1) if b1 is not None else None)
        hp.gemm(hd, wh1.data, g1, beta=1.0)
        z, r, rh = (hp.empty((B, H), np.float32) for _ in range(3))
        L.call("pdn_gru_gates_fwd_f32", g1._ptr, hd._ptr, z._ptr, r._ptr, rh._ptr, B, H, hp.stream())
        g2 = hp.empty((B, H), np.float32)
        hp.gemm(xd, wx2.data, g2, bias=b2.data.reshape(-1) if b2 is not None else None)
        hp.gemm(rh, wh2.data, g2, beta=1.0)
        n, hn = hp.empty((B, H), np.float32), hp.empty((B, H), np.float32)
        L.call("pdn_gru_out_fwd_f32", g2._ptr, z._ptr, hd._ptr, n._ptr, hn._ptr, B, H, hp.stream())
        self._saved = (xd, hd, z, r, rh, n)
        return hn

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, h, wx1, wh1, wx2, wh2 = self.last[:6]
        b1, b2 = (self.last[6], self.last[7]) if self.has_bias else (None, None)
        xd, hd, z, r, rh, n = self._saved
        B, H = hd.shape
        g = _contig(g)
        dg2, dg1, dh = hp.empty((B, H), np.float32), hp.empty((B, 2 * H), np.float32), hp.empty((B, H), np.float32)
        L.call("pdn_gru_out_bwd_f32", g._ptr, z._ptr, n._ptr, hd._ptr, dg2._ptr, dg1._ptr, dh._ptr, B, H, hp.stream())
        drh = hp.empty((B, H), np.float32)
        hp.gemm(dg2, wh2.data.T, drh)
        L.call("pdn_gru_gates_bwd_f32", drh._ptr, r._ptr, hd._ptr, dg1._ptr, dh._ptr, B, H, hp.stream())
        grads = [None] * len(self.last)
        if h.requires_grad:
            hp.gemm(dg1, wh1.data.T, dh, beta=1.0)
            grads[1] = dh
        if x.requires_grad:
            dx = hp.empty(x.shape, np.float32)
            hp.gemm(dg2, wx2.data.T, dx)
            hp.gemm(dg1, wx1.data.T, dx, beta=1.0)
            grads[0] = dx
        for idx, (a, d, w) in enumerate(((xd, dg1, wx1), (hd, dg1, wh1), (xd, dg2, wx2), (rh, dg2, wh2)), start=2):
            if not w.requires_grad:
                continue
            if _is_leaf_f32(w):
                hp.gemm(a.T, d, w.grad, beta=1.0)
            else:
                dw = hp.empty(w.shape, np.float32)
                hp.gemm(a.T, d, dw)
                grads[idx] = dw
        if self.has_bias:
            if b1.requires_grad:
                grads[6] = dg1.sum(0).reshape(b1.shape)
            if b2.requires_grad:
                grads[7] = dg2.sum(0).reshape(b2.shape)
        return grads


def _pack_columns(hp, weights):
    """(in, sum out_i) copy of weights that share `in`: the B operand of ONE input-gradient GEMM
    dX = [d_1 | d_2 | ...] @ [W_1 | W_2 | ...]^T with the contraction running over all projections at
    once (a few hundred KB per call: cheaper than accumulating K-split products through dX)."""
    fin = weights[0].shape[0]
    outs = [w.shape[1] for w in weights]
    cat = hp.empty((fin, int(np.sum(outs))), np.float32)
    pos = 0
    for w, n in zip(weights, outs):
        cat[:, pos:pos + n] = w
        pos += n
    return cat


def _dx_of_shared_input(hp, node, x, dcat, weights, T, fin):
    """dx = [d_1 | d_2 | ...] @ [W_1 | W_2 | ...]^T (+ the gradient x already holds) for projections that share their
    input: ONE contraction over all of them.  With the weights equally spaced in memory (how `Attention.move` /
    `FeedForward.move` pack them) and enough rows the product reads them where they live
    (`pdn_gemm_outres_blocks_nt_f32`); otherwise against a column-packed copy made here."""
    L = _L()
    dx = hp.empty(x.shape, np.float32)
    ex = _foldable(node, 0, x)
    exr = ex.reshape(T, fin) if ex is not None else None
    ws = [_contig(w.data) for w in weights]
    stack = hp.stacked_view(ws)
    kb = ws[0].shape[1]
    if (fin == 288 and stack is not None and abs(stack._strides[0]) < (1 << 40) and dcat.is_contiguous()
            and os.environ.get("PDN_NO_DX_BLOCKS", "0") != "1"
            and L.query("pdn_gemm_outres_blocks_supported", T, kb, len(ws))):
        L.call("pdn_gemm_outres_blocks_nt_f32", dcat._ptr, ws[0]._ptr, stack._strides[0], kb, len(ws),
               dx._ptr, exr._ptr if exr is not None else None, T, dcat.shape[1], fin, hp.stream())
    else:
        wcat = _pack_columns(hp, [w.data for w in weights])                    # (fin, sum out_i)
        hp.gemm(dcat, wcat.T, dx.reshape(T, fin), residual=exr)
    return dx


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
        x2 = _contig(x.data).reshape(T, D)
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
        if self.rotated:
            tab = self._rope_table(cos, sin, Lq, hd)
            L.call("pdn_qkv_rope_fwd_f32", x2._ptr, ws[0]._ptr, (ws[1]._ptr - ws[0]._ptr) // 4, qkv._ptr, tab._ptr,
                   T, D, D, Lq, hd, D, hp.stream())
        elif stack is not None:
            hp.gemm(x2, stack, blocks)
        else:
            for i in range(3):
                hp.gemm(x2, ws[i], blocks[i])
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


class gate_up_swiglu(_Operator):
    """The FFN front end as ONE tape node (llm/llama/model.py:56-58): h = silu(x Wg) * (x Wu).
    Both bias-free projections write the halves of ONE packed (tokens, 2 * ffn) buffer (a single batched
    GEMM when the two weights are equally spaced in memory), the SwiGLU kernel reads the halves through
    a row stride.  Backward: d[gate | up] into one packed buffer, the two weight gradients as one
    batched GEMM and dx as ONE GEMM contracting over all 2 * ffn columns (the reference: 2 matmul + 5
    elementwise nodes forward, 2 separately accumulated input gradients backward)."""

    folds_existing = True
    enabled = True

    @staticmethod
    def applicable(x, wg, wu):
        return (gate_up_swiglu.enabled and x.device.is_hip and x.dtype == np.float32 and wg.dtype == np.float32
                and wu.dtype == np.float32 and wg.shape == wu.shape and wg.shape[1] % 4 == 0 and x.ndim >= 2)

    def __init__(self, x, w_gate, w_up):
        super().__init__(x, w_gate, w_up)

    @staticmethod
    def _halves(buf, T, F):
        hp = _hip()
        return hp.ndarray(buf._buf, buf._ptr, (2, T, F), (F, 2 * F, 1), buf.dtype)

    def forward_(self, x, wg, wu):
        _require_f32(self, x, wg, wu)
        hp, L = _hip(), _L()
        fin, F = wg.shape
        x2 = _contig(x.data).reshape(-1, fin)
        T = x2.shape[0]
        gu = hp.empty((T, 2 * F), np.float32)
        halves = self._halves(gu, T, F)
        ws = [_contig(wg.data), _contig(wu.data)]
        stack = hp.stacked_view(ws)
        if stack is not None:
            hp.gemm(x2, stack, halves)
        else:
            hp.gemm(x2, ws[0], halves[0])
            hp.gemm(x2, ws[1], halves[1])
        out = hp.empty(x.shape[:-1] + (F,), np.float32)
        L.call("pdn_swiglu_rows_fwd_f32", gu._ptr, out._ptr, T, F, hp.stream())
        self._saved = (x2, gu)
        return out

    def backward_all(self, dh):
        hp, L = _hip(), _L()
        x, wg, wu = self.last
        fin, F = wg.shape
        x2, gu = self._saved
        T = x2.shape[0]
        dh = _contig(dh)
        dgu = hp.empty((T, 2 * F), np.float32)
        L.call("pdn_swiglu_rows_bwd_f32", gu._ptr, dh._ptr, dgu._ptr, T, F, hp.stream())
        dhalves = self._halves(dgu, T, F)
        grads = [None] * 3
        weights = (wg, wu)
        gstack = None
        if all(w.requires_grad and _is_leaf_f32(w) for w in weights):
            gstack = hp.stacked_view([w.grad for w in weights])
        side = _beside(hp, fin, 2 * F, x.requires_grad and any(w.requires_grad for w in weights))
        dws = [hp.empty(w.shape, np.float32) if gstack is None and w.requires_grad and not _is_leaf_f32(w) else None
               for w in weights]
        with side or contextlib.nullcontext():
            if gstack is not None:
                hp.gemm(x2.T, dhalves, gstack, beta=1.0)
            else:
                for i, w in enumerate(weights):
                    if not w.requires_grad:
                        continue
                    if dws[i] is None:
                        hp.gemm(x2.T, dhalves[i], w.grad, beta=1.0)
                    else:
                        hp.gemm(x2.T, dhalves[i], dws[i])
                        grads[1 + i] = dws[i]
        if x.requires_grad:
            dx = hp.empty(x.shape, np.float32)
            ex = _foldable(self, 0, x)
            wcat = _pack_columns(hp, [wg.data, wu.data])                           # (fin, 2F)
            hp.gemm(dgu, wcat.T, dx.reshape(T, fin), residual=ex.reshape(T, fin) if ex is not None else None)
            grads[0] = dx
        if side is not None:
            side.join()
        return grads


class ffn_swiglu(_Operator):
    """The whole feed-forward block as ONE tape node (llm/llama/model.py:47-58):
    y = (silu(x Wg) * (x Wu)) Wd (+ residual).  What the merge buys over `gate_up_swiglu` + `linear` is that the two
    bandwidth passes of SwiGLU ride in GEMM epilogues (csrc/gemm_rowres.hip, round 4): forward, the packed gate | up
    projection writes h = silu(gate) * up beside [gate | up] in the same launch; backward, dh = dy Wd^T is never
    written -- the product's store reads the saved gate / up and leaves d[gate | up].  Shapes the epilogue kernels do
    not take (contraction other than 288, ffn not a multiple of 96, few rows) run the same algebra with the separate
    SwiGLU kernels.  The reference: 3 matmul + 5 elementwise nodes forward and their per-edge gradients."""

    folds_existing = True
    enabled = os.environ.get("PDN_NO_FFN_NODE", "0") != "1"            # (same-box A/B switches)
    epilogues = os.environ.get("PDN_NO_SWIGLU_EPILOGUE", "0") != "1"
    epilogue_min_rows = 4096         # below this the projections are launch-sized: the separate kernels are as good

    @staticmethod
    def applicable(x, wg, wu, wd):
        return (ffn_swiglu.enabled and gate_up_swiglu.applicable(x, wg, wu) and wd.dtype == np.float32
                and wd.shape == (wg.shape[1], wg.shape[0]))

    def __init__(self, x, w_gate, w_up, w_down, residual=None):
        self.has_res = residual is not None
        super().__init__(*([x, w_gate, w_up, w_down] + ([residual] if self.has_res else [])))

    @staticmethod
    def _epilogue(T, F, fin, stack):
        # (two arrays are always "equally spaced": the kernel's 32-bit offsets need them within 2^30 floats, not overlapping)
        return bool(ffn_swiglu.epilogues and stack is not None and T >= ffn_swiglu.epilogue_min_rows
                    and fin * F <= abs(stack._strides[0]) < (1 << 30) - fin * F
                    and _L().query("pdn_gateup_swiglu_supported", T, F, fin))

    def forward_(self, x, wg, wu, wd, r=None):
        _require_f32(self, x, wg, wu, wd, r)
        hp, L = _hip(), _L()
        fin, F = wg.shape
        x2 = _contig(x.data).reshape(-1, fin)
        T = x2.shape[0]
        gu = hp.empty((T, 2 * F), np.float32)
        h = hp.empty((T, F), np.float32)
        ws = [_contig(wg.data), _contig(wu.data)]
        stack = hp.stacked_view(ws)
        self.used_epilogue = self._epilogue(T, F, fin, stack)
        if self.used_epilogue:
            L.call("pdn_gateup_swiglu_fwd_f32", x2._ptr, ws[0]._ptr, (ws[1]._ptr - ws[0]._ptr) // 4, gu._ptr, h._ptr,
                   T, F, fin, fin, hp.stream())
        else:
            halves = gate_up_swiglu._halves(gu, T, F)
            if stack is not None:
                hp.gemm(x2, stack, halves)
            else:
                hp.gemm(x2, ws[0], halves[0])
                hp.gemm(x2, ws[1], halves[1])
            L.call("pdn_swiglu_rows_fwd_f32", gu._ptr, h._ptr, T, F, hp.stream())
        out = hp.empty(x.shape[:-1] + (fin,), np.float32)
        res = _contig(r.data).reshape(-1, fin) if r is not None else None
        hp.gemm(h, wd.data, out.reshape(-1, fin), residual=res)
        self._saved = (x2, gu, h)
        return out

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, wg, wu, wd = self.last[:4]
        fin, F = wg.shape
        x2, gu, h = self._saved
        T = x2.shape[0]
        grads = [None] * len(self.last)
        if self.has_res and self.last[4].requires_grad:
            grads[4] = g
        g2 = _contig(g).reshape(T, fin)
        # down projection: dWd += h^T g
        if wd.requires_grad:
            if _is_leaf_f32(wd):
                hp.gemm(h.T, g2, wd.grad, beta=1.0)
            else:
                grads[3] = hp.empty((F, fin), np.float32)
                hp.gemm(h.T, g2, grads[3])
        if not (x.requires_grad or wg.requires_grad or wu.requires_grad):
            return grads
        # d[gate | up] = SwiGLU'(gate, up) o (g Wd^T)
        dgu = hp.empty((T, 2 * F), np.float32)
        wdd = _contig(wd.data)
        if self.used_epilogue:
            L.call("pdn_swiglu_bwd_gemm_f32", g2._ptr, wdd._ptr, gu._ptr, dgu._ptr, T, F, fin, fin, hp.stream())
        else:
            dh = hp.empty((T, F), np.float32)
            hp.gemm(g2, wdd.T, dh)
            L.call("pdn_swiglu_rows_bwd_f32", gu._ptr, dh._ptr, dgu._ptr, T, F, hp.stream())
        dhalves = gate_up_swiglu._halves(dgu, T, F)
        weights = (wg, wu)
        gstack = None
        if all(w.requires_grad and _is_leaf_f32(w) for w in weights):
            gstack = hp.stacked_view([w.grad for w in weights])
        if gstack is not None:
            hp.gemm(x2.T, dhalves, gstack, beta=1.0)
        else:
            for i, w in enumerate(weights):
                if not w.requires_grad:
                    continue
                if _is_leaf_f32(w):
                    hp.gemm(x2.T, dhalves[i], w.grad, beta=1.0)
                else:
                    grads[1 + i] = hp.empty(w.shape, np.float32)
                    hp.gemm(x2.T, dhalves[i], grads[1 + i])
        if x.requires_grad:
            # (a residual that IS x hands its gradient g over separately: the engine adds it)
            grads[0] = _dx_of_shared_input(hp, self, x, dgu, (wg, wu), T, fin)
        return grads


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


def _gemm_raw(L, st, M, N, K, a_ptr, a_rs, a_cs, b, c_ptr, ldc, beta=0.0, residual_ptr=None, b_transposed=False):
    """pdn_gemm_f32 on raw pointers (time loops: skips the per-call view / workspace bookkeeping of
    hipnp.gemm); `b` is a 2-D hipnp array, used as b or b.T."""
    rs, cs = (b._strides[1], b._strides[0]) if b_transposed else (b._strides[0], b._strides[1])
    L.call("pdn_gemm_f32", M, N, K, 1.0, a_ptr, a_rs, a_cs, b._ptr, rs, cs, beta, c_ptr, ldc, None, 1, 1,
           0, 0, 0, 0, 0, 0, residual_ptr, None, 0, None, 0, st)


class gru_sequence(_Operator):
    """A whole single-layer GRU over T steps as ONE tape node (nn/modules/rnn.py:537-544, 640-694):
    the input projections of all steps are hoisted into two GEMMs over (T*B, in); each step then costs
    two (B, H) x (H, .) GEMMs (the hoisted term rides in as the epilogue residual) and the two gate
    kernels; backward walks the steps in reverse with five launches each and forms every weight
    gradient with ONE long-K GEMM over the stacked per-step quantities.  The reference runs ~20 tape
    nodes per step.  Inputs: x (T, B, in), h0 (B, H), Wx1, Wh1, Wx2, Wh2[, b1, b2]; output (T, B, H)."""

    use_persistent = True      # class switch: False keeps the per-step launches (tests, A/B)

    def __init__(self, x, h0, wx1, wh1, wx2, wh2, b1=None, b2=None):
        self.has_bias = b1 is not None
        super().__init__(*((x, h0, wx1, wh1, wx2, wh2) + ((b1, b2) if self.has_bias else ())))

    def forward_(self, x, h0, wx1, wh1, wx2, wh2, b1=None, b2=None):
        if self.xp is np:
            raise NotImplementedError("gru_sequence is the HIP fused path")
        _require_f32(self, x, h0, wx1, wh1, wx2, wh2, b1, b2)
        hp, L = _hip(), _L()
        T, B, I = x.shape
        H = h0.shape[-1]
        x2 = _contig(x.data).reshape(T * B, I)
        g1x, g2x = hp.empty((T, B, 2 * H), np.float32), hp.empty((T, B, H), np.float32)
        hp.gemm(x2, wx1.data, g1x.reshape(T * B, 2 * H), bias=b1.data.reshape(-1) if b1 is not None else None)
        hp.gemm(x2, wx2.data, g2x.reshape(T * B, H), bias=b2.data.reshape(-1) if b2 is not None else None)
        out = hp.empty((T, B, H), np.float32)
        Z, R, RH, N = (hp.empty((T, B, H), np.float32) for _ in range(4))
        h0d = _contig(h0.data)
        st = hp.stream()
        self._persistent = bool(gru_sequence.use_persistent and L.query("pdn_gru_seq_supported", H))
        if self._persistent:
            # the whole time loop in ONE launch: a wave owns 32 sequences, h stays in its registers
            L.call("pdn_gru_seq_fwd_f32", g1x._ptr, g2x._ptr, h0d._ptr, _contig(wh1.data)._ptr, _contig(wh2.data)._ptr,
                   Z._ptr, R._ptr, RH._ptr, N._ptr, out._ptr, T, B, H, st)
            self._saved = (x2, h0d, Z, R, RH, N)
            return out
        g1, g2 = hp.empty((B, 2 * H), np.float32), hp.empty((B, H), np.float32)
        wh1d, wh2d = wh1.data, wh2.data
        sH, s2H = B * H * 4, B * 2 * H * 4                           # bytes per time step
        hprev = h0d._ptr
        for t in range(T):
            z, r, rh, n, o = Z._ptr + t * sH, R._ptr + t * sH, RH._ptr + t * sH, N._ptr + t * sH, out._ptr + t * sH
            _gemm_raw(L, st, B, 2 * H, H, hprev, H, 1, wh1d, g1._ptr, 2 * H, residual_ptr=g1x._ptr + t * s2H)
            L.call("pdn_gru_gates_fwd_f32", g1._ptr, hprev, z, r, rh, B, H, st)
            _gemm_raw(L, st, B, H, H, rh, H, 1, wh2d, g2._ptr, H, residual_ptr=g2x._ptr + t * sH)
            L.call("pdn_gru_out_fwd_f32", g2._ptr, z, hprev, n, o, B, H, st)
            hprev = o
        self._saved = (x2, h0d, Z, R, RH, N)
        return out

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, h0, wx1, wh1, wx2, wh2 = self.last[:6]
        b1, b2 = (self.last[6], self.last[7]) if self.has_bias else (None, None)
        x2, h0d, Z, R, RH, N = self._saved
        T, B, H = Z.shape
        I = x2.shape[1]
        g = _contig(g)
        out, st = self.data, hp.stream()
        dG1, dG2 = hp.empty((T, B, 2 * H), np.float32), hp.empty((T, B, H), np.float32)
        if self._persistent:
            dh = hp.empty((B, H), np.float32)
            L.call("pdn_gru_seq_bwd_f32", g._ptr, Z._ptr, R._ptr, N._ptr, out._ptr, h0d._ptr, _contig(wh1.data)._ptr,
                   _contig(wh2.data)._ptr, dG1._ptr, dG2._ptr, dh._ptr, T, B, H, st)
        else:
            dh = self._backward_steps(hp, L, st, g, out, h0d, Z, R, N, dG1, dG2, wh1.data, wh2.data, T, B, H)
        return self._finish_backward(hp, x, h0, wx1, wh1, wx2, wh2, b1, b2, x2, h0d, out, RH, dG1, dG2, dh, T, B, H, I)

    @staticmethod
    def _backward_steps(hp, L, st, g, out, h0d, Z, R, N, dG1, dG2, wh1d, wh2d, T, B, H):
        dh, dh2, drh = hp.zeros((B, H), np.float32), hp.empty((B, H), np.float32), hp.empty((B, H), np.float32)
        sH, s2H = B * H * 4, B * 2 * H * 4
        for t in range(T - 1, -1, -1):
            hprev = out._ptr + (t - 1) * sH if t > 0 else h0d._ptr
            z, r, n = Z._ptr + t * sH, R._ptr + t * sH, N._ptr + t * sH
            dg1, dg2 = dG1._ptr + t * s2H, dG2._ptr + t * sH
            # dh += g[t]: gradient of h_t = direct + from step t+1
            dh += g[t]
            L.call("pdn_gru_out_bwd_f32", dh._ptr, z, n, hprev, dg2, dg1, dh2._ptr, B, H, st)
            _gemm_raw(L, st, B, H, H, dg2, H, 1, wh2d, drh._ptr, H, b_transposed=True)
            L.call("pdn_gru_gates_bwd_f32", drh._ptr, r, hprev, dg1, dh2._ptr, B, H, st)
            _gemm_raw(L, st, B, H, 2 * H, dg1, 2 * H, 1, wh1d, dh2._ptr, H, beta=1.0, b_transposed=True)
            dh, dh2 = dh2, dh
        return dh

    def _finish_backward(self, hp, x, h0, wx1, wh1, wx2, wh2, b1, b2, x2, h0d, out, RH, dG1, dG2, dh, T, B, H, I):
        grads = [None] * len(self.last)
        if h0.requires_grad:
            grads[1] = dh
        d1, d2 = dG1.reshape(T * B, 2 * H), dG2.reshape(T * B, H)
        if x.requires_grad:
            dx = hp.empty(x.shape, np.float32)
            hp.gemm(d1, wx1.data.T, dx.reshape(T * B, I))
            hp.gemm(d2, wx2.data.T, dx.reshape(T * B, I), beta=1.0)
            grads[0] = dx
        hprev_all = hp.empty((T, B, H), np.float32)                      # h_{t-1} for every step, stacked
        hprev_all[0] = h0d
        if T > 1:
            hprev_all[1:] = out[:T - 1]
        stacked = ((x2, d1, wx1), (hprev_all.reshape(T * B, H), d1, wh1), (x2, d2, wx2), (RH.reshape(T * B, H), d2, wh2))
        for idx, (a, d, w) in enumerate(stacked, start=2):
            if not w.requires_grad:
                continue
            if _is_leaf_f32(w):
                hp.gemm(a.T, d, w.grad, beta=1.0)
            else:
                dw = hp.empty(w.shape, np.float32)
                hp.gemm(a.T, d, dw)
                grads[idx] = dw
        if self.has_bias:
            if b1.requires_grad:
                grads[6] = d1.sum(0).reshape(b1.shape)
            if b2.requires_grad:
                grads[7] = d2.sum(0).reshape(b2.shape)
        return grads
