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

from ..tensor import (Graph, Tensor, _Operator, transpose, matmul as _matmul, div as _div, add as _add, sub as _sub, mul as _mul,
                      reshape as _reshape, concat as _concat, _get_slice)
from ._common import _Deferred
from .attn import attention, _attn_layout
from .dense import linear as _linear, linear_cross_entropy as _linear_ce
from .pointwise import softmax as _softmax, rope as _rope, silu as _silu, swiglu as _swiglu


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
    """Hook of Tensor.reshape: a pending projection whose LAST axis is kept; the tail of the rotary embedding."""
    if type(t) is rope_link:
        return _rope_reshape(t, new_shape)
    if type(t) is _concat:
        return _rope_tail(t, new_shape)
    if type(t) is not _linear:
        return _rope_pairs(t, new_shape)
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


# ---- the backward pass of the reference's rotary embedding (llm/llama/model.py:23-44) ----------------------------------
#
#     xri = x.reshape(*x.shape[:-1], -1, 2);  r, i = xri[..., 0], xri[..., 1]
#     c, s = unsqueeze(cos, -2), unsqueeze(sin, -2)
#     out = concat([unsqueeze(r * c - i * s, -1), unsqueeze(r * s + i * c, -1)], -1).reshape(*x.shape)
#
# thirteen nodes per rotated tensor whose BACKWARD is what costs: four strided products, two scatter-assigns of a slice into
# zeros, their sums, a negation (~270 us per tensor at 65536 tokens against ~35 us for the rotation by -theta that they
# amount to).  The forward values are kept exactly as the plain operators produced them; when the final `reshape` finds
# precisely this expression over ONE tensor x and two tables without gradient, its result is put on the tape as a single
# node over x whose gradient is `fused.rope`'s (rotation of the upstream gradient by -theta): same mathematics, rounding
# order of one multiply-add pair per element aside.  Anything else -- a different expression, tables that need a gradient --
# is left alone.  An intermediate that has OTHER consumers keeps working: the thirteen nodes stay on the tape and carry
# whatever gradient those consumers send (gradients add over paths); only the path through this result runs as one node.
class rope_chain:
    enabled = True        # class switch: False keeps the thirteen nodes on the tape (A/B and tests)
    taken = 0             # expressions whose backward became one node


class rope_taken(_rope):
    """`fused.rope` over values somebody else already computed (the plain-operator expression above)."""

    def __init__(self, x, cos, sin, data):
        self._given = data
        super().__init__(x, cos, sin)

    def forward_(self, x):
        d, self._given = self._given, None
        return d


def _unsqueezed(t, axis_from_end):
    """x if t is `reshape(x, shape with a 1 inserted at -axis_from_end)` (pdn.unsqueeze), else None."""
    if type(t) is not _reshape:
        return None
    x = getattr(t, "_src", None)
    if x is None:
        return None
    want = list(x.shape)
    want.insert(len(want) + 1 - axis_from_end, 1)
    return x if tuple(t.shape) == tuple(want) else None


def _pair_slice(t, which):
    """pairs if t is `pairs[..., which]`, else None."""
    if type(t) is _get_slice and isinstance(t.key, tuple) and len(t.key) == 2 and t.key[0] is Ellipsis and \
            isinstance(t.key[1], (int, np.integer)) and int(t.key[1]) == which and t.last:
        return t.last[0]
    return None


def _product(t):
    return tuple(t.last) if type(t) is _mul and len(t.last) == 2 else None


def _rope_tail(cat, new_shape):
    if not (rope_chain.enabled and cat.requires_grad and len(cat.tensors) == 2 and cat.ndim == 5
            and cat.axis in (-1, 4) and cat.device.is_hip and cat.dtype == np.float32):
        return None
    A, B = _unsqueezed(cat.tensors[0], 1), _unsqueezed(cat.tensors[1], 1)
    if A is None or B is None or type(A) is not _sub or type(B) is not _add:
        return None
    pa, pb, pc, pd = _product(A.last[0]), _product(A.last[1]), _product(B.last[0]), _product(B.last[1])
    if None in (pa, pb, pc, pd):
        return None

    def split(p):                        # (slice of pairs, table view) in either order
        for u, v in (p, p[::-1]):
            if type(u) is _get_slice:
                return u, v
        return None, None
    (r1, c1), (i1, s1), (r2, s2), (i2, c2) = split(pa), split(pb), split(pc), split(pd)
    if r1 is None or r1 is not r2 or i1 is None or i1 is not i2 or c1 is not c2 or s1 is not s2:
        return None
    pairs = _pair_slice(r1, 0)
    if pairs is None or _pair_slice(i1, 1) is not pairs or type(pairs) is not _reshape or not pairs.last:
        return None
    x = pairs.last[0]
    cos, sin = _unsqueezed(c1, 2), _unsqueezed(s1, 2)
    if cos is None or sin is None or cos.requires_grad or sin.requires_grad or x.ndim != 4 or not x.requires_grad:
        return None
    Bq, Lq, H, hd = x.shape
    if hd % 2 or tuple(pairs.shape) != (Bq, Lq, H, hd // 2, 2) or tuple(cos.shape) != (Lq, hd // 2) or \
            tuple(sin.shape) != (Lq, hd // 2) or cos.dtype != np.float32 or sin.dtype != np.float32:
        return None
    if len(new_shape) == 1 and isinstance(new_shape[0], (tuple, list)):
        new_shape = tuple(new_shape[0])
    shape = [int(v) for v in new_shape]
    if len(shape) != 4 or shape[:3] != [Bq, Lq, H] or shape[3] not in (-1, hd):
        return None
    rope_chain.taken += 1
    return rope_taken(x, cos, sin, cat.data.reshape(Bq, Lq, H, hd))


# ---- silu(gate) * up (llm/llama/model.py:56-58 with nn/functional.py:39-40) --------------------------------------------
# `F.silu` of a HIP float32 tensor is a pending node; a product with a tensor of the same shape takes it over as ONE
# `swiglu` node (one pass forward, one backward, instead of the activation, a product and its two gradient products).
class swiglu_chain:
    enabled = True
    taken = 0


def on_mul(a, b):
    """Hook of Tensor.__mul__ when one factor is a pending `silu`."""
    if not (isinstance(a, Tensor) and isinstance(b, Tensor)):
        return None
    if type(a) is rope_link or type(b) is rope_link:
        return _rope_product(a, b)
    if not swiglu_chain.enabled:
        return None
    for act, other in ((a, b), (b, a)):
        if type(act) is _silu and act._pending is not None and other is not act and \
                tuple(other.shape) == tuple(act.shape) and other.dtype == np.float32 and other.device == act.device:
            swiglu_chain.taken += 1
            return _swiglu(act._pending[0], other)
    return None


# ---- the rotary embedding written with plain operators, FORWARD too (llm/llama/model.py:23-44) -------------------------
# Every operator of the expression documented above `rope_chain` is a pending link while nobody reads it:
#     pairs = x.reshape(..., hd/2, 2) -> comp = pairs[..., 0 | 1] -> prod = comp * unsqueeze(table, -2)
#     -> diff = prod - prod, sum = prod + prod -> unsq = unsqueeze(., -1) -> cat = concat([unsq, unsq], -1) -> reshape(x.shape)
# and the last reshape, finding exactly `concat([r c - i s, r s + i c])` over ONE x and two tables without gradient, builds ONE
# `fused.rope` node (one kernel forward, one backward) instead of thirteen.  A link somebody reads builds its ordinary
# operator then (over the previous link, which does the same) and becomes an identity over it, with a fresh place in the
# tape -- programs that look at the pairs or the products behave as before.
class rope_link(_Deferred, _Operator):
    _rope_link = True
    _mul_hook = True
    _reshape_hook = True
    enabled = True        # class switch: False runs the plain operators (the backward-only node of `rope_chain` still applies)
    fused_built = 0       # expressions that became one fused.rope node

    def __init__(self, stage, inputs, shape, **info):
        self.stage, self.info = stage, info
        self._init_deferred(inputs, shape, np.float32)

    def forward_(self, *ins):
        st = self.stage
        if st == "pairs" or st == "unsq":
            inner = _reshape(ins[0], self._shape)
        elif st == "comp":
            inner = _get_slice(ins[0], (Ellipsis, self.info["which"]))
        elif st == "prod":
            inner = _mul(ins[0], ins[1])
        elif st == "diff":
            inner = _sub(ins[0], ins[1])
        elif st == "sum":
            inner = _add(ins[0], ins[1])
        else:                                            # "cat" (constructed past concat.__new__'s own hook)
            inner = object.__new__(_concat)
            inner.__init__(list(ins), axis=-1)
        if self.requires_grad and inner.requires_grad:
            self.last = [inner]                          # (see attn_link.forward_: a fresh place in the registry)
            Graph._free_node(self)
            self.last = [inner]
            Graph._add_node(self)
        return inner.data

    def grad_fn(self, x, grad):
        return grad


def _link(t, stage):
    return type(t) is rope_link and t._pending is not None and t.stage == stage


def _resolve(shape, size):
    shape = [int(v) for v in shape]
    if shape.count(-1) == 1:
        known = int(np.prod([v for v in shape if v != -1], dtype=np.int64))
        if known == 0 or size % known:
            return None
        shape[shape.index(-1)] = size // known
    return shape if all(v >= 0 for v in shape) and int(np.prod(shape, dtype=np.int64)) == size else None


def _rope_pairs(x, new_shape):
    """x.reshape(*x.shape[:-1], -1, 2) of a 4-D float32 HIP tensor."""
    if not (rope_link.enabled and x.ndim == 4 and x.device.is_hip and x.dtype == np.float32):
        return None
    if len(new_shape) == 1 and isinstance(new_shape[0], (tuple, list)):
        new_shape = tuple(new_shape[0])
    if len(new_shape) != 5 or new_shape[-1] != 2:
        return None
    shape = _resolve(new_shape, x.size)
    if shape is None or tuple(shape[:3]) != tuple(x.shape[:3]) or x.shape[3] % 2 or shape[3] != x.shape[3] // 2:
        return None
    return rope_link("pairs", (x,), shape)


def on_getitem(t, key):
    """Hook of Tensor.__getitem__ on a pending link: pairs[..., 0] / pairs[..., 1]."""
    if _link(t, "pairs") and isinstance(key, tuple) and len(key) == 2 and key[0] is Ellipsis and \
            isinstance(key[1], (int, np.integer)) and int(key[1]) in (0, 1):
        return rope_link("comp", (t,), t.shape[:-1], which=int(key[1]))
    return None


def _rope_product(a, b):
    for comp, view in ((a, b), (b, a)):
        if _link(comp, "comp") and type(view) is _reshape:
            table = _unsqueezed(view, 2)
            L, half = comp.shape[1], comp.shape[3]
            if table is not None and not table.requires_grad and tuple(table.shape) == (L, half) and \
                    table.dtype == np.float32 and table.device == comp.device:
                return rope_link("prod", (a, b), comp.shape, comp=comp, table=table)
    return None


def on_addsub(a, b, op):
    """Hook of Tensor.__sub__ / __add__ on a pending link: prod - prod, prod + prod."""
    if _link(a, "prod") and _link(b, "prod") and a is not b:
        return rope_link("diff" if op == "sub" else "sum", (a, b), a.shape)
    return None


def _rope_reshape(t, new_shape):
    if len(new_shape) == 1 and isinstance(new_shape[0], (tuple, list)):
        new_shape = tuple(new_shape[0])
    if (_link(t, "diff") or _link(t, "sum")) and len(new_shape) == 5:            # pdn.unsqueeze(., -1)
        shape = _resolve(new_shape, t.size)
        if shape is not None and tuple(shape) == tuple(t.shape) + (1,):
            return rope_link("unsq", (t,), shape)
        return None
    if not (_link(t, "cat") and len(new_shape) == 4):
        return None
    shape = _resolve(new_shape, t.size)
    u_re, u_im = t._pending
    d, a = u_re._pending[0], u_im._pending[0]                                     # (all links of a pending cat are pending)
    if shape is None or d.stage != "diff" or a.stage != "sum":
        return None
    (rc, is_), (rs, ic) = d._pending, a._pending
    r, i = rc.info["comp"], is_.info["comp"]
    cos, sin = rc.info["table"], is_.info["table"]
    if not (rs.info["comp"] is r and ic.info["comp"] is i and rs.info["table"] is sin and ic.info["table"] is cos and
            cos is not sin and r.info["which"] == 0 and i.info["which"] == 1 and r._pending[0] is i._pending[0]):
        return None
    x = r._pending[0]._pending[0]
    if tuple(shape) != tuple(x.shape):
        return None
    rope_link.fused_built += 1
    return _rope(x, cos, sin)


def on_concat(tensors, axis):
    """Hook of concat.__new__: two pending unsqueezed combinations joined on the last axis."""
    tensors = list(tensors)
    if len(tensors) == 2 and _link(tensors[0], "unsq") and _link(tensors[1], "unsq") and tensors[0] is not tensors[1] and \
            axis in (-1, 4) and tuple(tensors[0].shape) == tuple(tensors[1].shape):
        links = [tensors[0], tensors[1], tensors[0]._pending[0], tensors[1]._pending[0]]
        links += list(links[2]._pending) + list(links[3]._pending)
        if all(type(k) is rope_link and k._pending is not None for k in links):
            return rope_link("cat", tuple(tensors), tuple(tensors[0].shape[:-1]) + (2,))
    return None

