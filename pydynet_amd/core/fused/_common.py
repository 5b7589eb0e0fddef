"""Helpers shared by the fused tape nodes: library handles, dtype guards, gradient folding, the deferred-node mixin.
(One module of `pydynet_amd.core.fused`; the package docstring lists the reference chains each node replaces.)"""
from __future__ import annotations

import os

import numpy as np

from ...autograd import is_grad_enable
from ..tensor import Graph


def _hip():
    from ... import hipnp
    return hipnp


def _L():
    from ... import _lib
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


def _gemm_raw(L, st, M, N, K, a_ptr, a_rs, a_cs, b, c_ptr, ldc, beta=0.0, residual_ptr=None, b_transposed=False):
    """pdn_gemm_f32 on raw pointers (time loops: skips the per-call view / workspace bookkeeping of
    hipnp.gemm); `b` is a 2-D hipnp array, used as b or b.T."""
    rs, cs = (b._strides[1], b._strides[0]) if b_transposed else (b._strides[0], b._strides[1])
    L.call("pdn_gemm_f32", M, N, K, 1.0, a_ptr, a_rs, a_cs, b._ptr, rs, cs, beta, c_ptr, ldc, None, 1, 1,
           0, 0, 0, 0, 0, 0, residual_ptr, None, 0, None, 0, st)
