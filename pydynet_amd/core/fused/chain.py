"""The reference's attention chain recognised as it is BUILT from plain operators (llm/llama/model.py:112-121):

    scores = q.transpose(0, 2, 1, 3) @ k.transpose(0, 2, 3, 1) / math.sqrt(hd)      # matmul, div by a host scalar
    scores = scores + mask                                                           # optional additive mask
    out    = F.softmax(scores, axis=-1) @ v.transpose(0, 2, 1, 3)                    # softmax, matmul

A program that keeps the reference's own model code (tests/models_plain_llama.py) builds exactly these nodes.  On a HIP
device every link of the chain is a DEFERRED node (`_Deferred`: shape and dtype answered, nothing run); when the last
matmul arrives with all links still pending the whole chain becomes ONE `fused.attention` node over q, k, v -- the same
mathematics (masked probabilities are exactly 0), nothing of size L x L in HBM -- and the links are dropped.  Any other
consumer of a link reads `.data`, which builds the ordinary operator then (the link turns into an identity over it), so
programs that look at the scores or the probabilities behave as before.  The additive mask is passed on as it is, except
when it is EXACTLY the causal mask of model.py:199-203 (found on the host when the Tensor is built from its NumPy
array, `Tensor._causal_mask`): then the kernels' own causal schedule is used and masked tiles are skipped."""
from __future__ import annotations

import math

import numpy as np

from ..tensor import Graph, Tensor, _Operator, transpose, matmul as _matmul, div as _div, add as _add
from ._common import _Deferred
from .attn import attention, _attn_layout
from .dense import linear as _linear, linear_cross_entropy as _linear_ce
from .pointwise import softmax as _softmax


class attn_link(_Deferred, _Operator):
    """One pending link (`stage` = "qk" | "scaled" | "masked" | "soft") of the chain above."""

    _pending_link = True
    enabled = True        # class switch: False keeps the plain operators (A/B and tests)
    fused_built = 0       # chains that became a fused.attention node (tests / bench assert on it)

    def __init__(self, stage, inputs, q, k, operand=None, mask=None):
        self.stage, self.q, self.k, self.operand, self.mask = stage, q, k, operand, mask
        B, Lq, H, _ = q.shape
        self._init_deferred(inputs, (B, H, Lq, k.shape[1]), np.float32)

    def forward_(self, *ins):
        # somebody wants this link's array: build the ordinary operator over the (then materialised) previous link
        if self.stage == "qk":
            inner = _matmul(ins[0], ins[1])
        elif self.stage == "scaled":
            inner = _div(ins[0], self.operand)
        elif self.stage == "masked":
            inner = _add(ins[0], ins[1])
        else:
            inner = _softmax(ins[0])
        if self.requires_grad and inner.requires_grad:
            # the engine walks ancestors in reverse CREATION order: this node now stands behind `inner`, which was
            # created after it -- it takes a fresh place in the registry
            self.last = [inner]
            Graph._free_node(self)
            self.last = [inner]
            Graph._add_node(self)
        return inner.data

    def grad_fn(self, x, grad):
        return grad


def _src(t, axes):
    """The tensor `t` is a `transpose(., axes)` view of, or None."""
    if type(t) is transpose and t.axes is not None and tuple(t.axes) == axes:
        return getattr(t, "_src", None)
    return None


def _pending(t, stages):
    return type(t) is attn_link and t._pending is not None and t.stage in stages


def on_matmul(a, b):
    """Hook of Tensor.__matmul__: a deferred link, a fused attention node (as its (B, H, L, hd) view), or None."""
    if not attn_link.enabled:
        return None
    if _pending(a, ("soft",)):
        v = _src(b, (0, 2, 1, 3))
        q, k = a.q, a.k
        if (v is not None and v.device == q.device and v.dtype == np.float32 and v.ndim == 4 and v.shape == k.shape
                and _attn_layout(v.data) is not None):
            mask = a.mask
            causal = bool(mask is not None and getattr(mask, "_causal_mask", False)
                          and tuple(mask.shape) == (q.shape[1], k.shape[1]))
            node = attention(q, k, v, causal=causal, start_pos=0, mask=None if (causal or mask is None) else mask)
            attn_link.fused_built += 1
            return transpose(node, (0, 2, 1, 3))
        return None
    q, k = _src(a, (0, 2, 1, 3)), _src(b, (0, 2, 3, 1))
    if q is None or k is None or not q.device.is_hip or q.device != k.device:
        return None
    if q.dtype != np.float32 or k.dtype != np.float32 or q.ndim != 4 or k.ndim != 4:
        return None
    if q.shape[0] != k.shape[0] or q.shape[2:] != k.shape[2:] or not attention.use_flash:
        return None
    if _attn_layout(q.data) is None or _attn_layout(k.data) is None:
        return None
    return attn_link("qk", (a, b), q, k)


def on_div(x, c):
    """Hook of Tensor.__truediv__: scores / sqrt(hd) with a host scalar."""
    if _pending(x, ("qk",)) and isinstance(c, (int, float, np.floating, np.integer)):
        hd = x.q.shape[3]
        if abs(float(c) - math.sqrt(hd)) <= 1e-6 * math.sqrt(hd):
            return attn_link("scaled", (x,), x.q, x.k, operand=c)
    return None


def on_add(x, m):
    """Hook of Tensor.__add__: scores + additive mask (a constant tensor broadcastable to (B, H, Lq, Lk))."""
    if _pending(x, ("scaled",)) and isinstance(m, Tensor) and not m.requires_grad and m.device == x.device \
            and m.dtype == np.float32 and m.ndim <= 4:
        want = x.shape
        shp = (1,) * (4 - m.ndim) + tuple(m.shape)
        if all(s in (1, t) for s, t in zip(shp, want)):
            return attn_link("masked", (x, m), x.q, x.k, mask=m)
    return None


def on_softmax(x):
    """Hook of F.softmax(., axis=-1)."""
    if _pending(x, ("scaled", "masked")):
        return attn_link("soft", (x,), x.q, x.k, mask=x.mask)
    return None


# ---- the loss of the reference's own training step (llm/llama/model.py:239-249) ---------------------------------------
#
#     logits = self.forward_logits(ids)                  # ... -> nn.Linear(dim, vocab)            (model.py:179)
#     loss = criterion(logits.reshape(B * L, V), targets)                                          (model.py:242-249)
#
# A projection the device could hand to a fused consumer is created without running (`fused.linear`, `_Deferred`).  While
# it is still pending, `reshape` that only regroups its leading axes is the same projection of the regrouped input, and
# `cross_entropy_loss` of it is ONE `linear_cross_entropy` node over (x, W, b): the (tokens x vocab) logits are written
# once, their gradient never exists, and both backward products form it from the saved logits -- the node
# `pydynet_amd.llm.llama.Llama.loss` builds by name.  The pending nodes that were passed over stay what they were: anybody
# who reads `logits` (an accuracy, a second loss term) gets the ordinary product then, on the tape as before.
class loss_chain:
    enabled = True        # class switch: False keeps linear -> reshape -> cross entropy as three nodes (A/B and tests)
    fused_built = 0       # chains that became a linear_cross_entropy node


def _pending_linear(t):
    return type(t) is _linear and t._pending is not None and not t.has_res


def on_reshape(t, new_shape):
    """Hook of Tensor.reshape: a pending projection whose LAST axis is kept."""
    if not (loss_chain.enabled and _pending_linear(t)):
        return None
    if len(new_shape) == 1 and isinstance(new_shape[0], (tuple, list)):
        new_shape = tuple(new_shape[0])
    shape = [int(v) for v in new_shape]
    if not shape or shape[-1] != t.shape[-1] or any(v == 0 for v in shape):
        return None
    rows = t.size // t.shape[-1]
    if shape.count(-1) == 1:
        known = int(np.prod([v for v in shape[:-1] if v != -1], dtype=np.int64))
        if known == 0 or rows % known:
            return None
        shape[shape.index(-1)] = rows // known
    if any(v < 0 for v in shape) or int(np.prod(shape[:-1], dtype=np.int64)) != rows:
        return None
    x, w = t._pending[0], t._pending[1]
    b = t._pending[2] if t.has_bias else None
    return _linear(x.reshape(*shape[:-1], x.shape[-1]), w, b)


def on_cross_entropy(y_pred, y_true, reduction):
    """Hook of F.cross_entropy_loss (class-index targets, 2-D float32 predictions)."""
    if not (loss_chain.enabled and _pending_linear(y_pred) and y_pred.ndim == 2):
        return None
    x, w = y_pred._pending[0], y_pred._pending[1]
    b = y_pred._pending[2] if y_pred.has_bias else None
    if not _linear_ce.applicable(x, w, b, y_true, reduction):
        return None
    loss_chain.fused_built += 1
    return _linear_ce(x, w, b, y_true, reduction)

