"""ORACLE (test infrastructure, never shipped on the product path).

A compact NumPy restatement of the reference's autograd semantics for the Tensor-op hot
path: what `pydynet/core/tensor.py` computes, forward and backward, op by op.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this
package; `pydynet_amd` never does.

Pinning: `tools/gen_golden.py` imports the real reference from /root/reference in the build
container and writes input/output vectors to `tests/golden/`; `tests/test_oracle_cpu.py`
checks this restatement against those vectors (and, when /root/reference is present,
against the live reference) -- bit-exact for index/reshape/im2col/gather paths, and
bit-exact or <= 1e-6 relative for float paths (same NumPy calls in the same order).

Structure is deliberately different from the reference (a functional tape of closures
instead of one Tensor subclass per operator); every rule cites the reference lines it
restates.  Citations are into /root/reference/pydynet/core/tensor.py unless noted.
"""
from __future__ import annotations

import numpy as np


class _State:
    grad_enabled = True   # autograd.py:3  (process-global flag)
    nodes: list = []      # Graph.node_list, tensor.py:11
    counter = 0


class no_grad:
    """autograd.py:15-30"""

    def __enter__(self):
        self.prev = _State.grad_enabled
        _State.grad_enabled = False

    def __exit__(self, *exc):
        _State.grad_enabled = self.prev


def set_grad_enabled(mode: bool):
    _State.grad_enabled = bool(mode)


def reset_tape():
    _State.nodes.clear()


class Var:
    """A value on the tape.  `parents` = [(Var, vjp)] where vjp(g) -> grad wrt that parent."""

    __slots__ = ("value", "requires_grad", "grad", "parents", "idx", "name")

    def __init__(self, value, dtype=None, requires_grad=False, _parents=None):
        if isinstance(value, Var):
            raise ValueError("Tensor assignment with another tensor is forbidden.")  # :73-75
        self.value = np.array(value, dtype=dtype) if not isinstance(value, np.ndarray) or dtype is not None \
            else value
        self.requires_grad = bool(_State.grad_enabled and requires_grad)             # :84
        self.parents = _parents or []
        self.grad = None
        self.name = None
        if self.requires_grad:
            if not np.issubdtype(self.value.dtype, np.floating):
                raise TypeError("Only Tensors of floating point dtype can require gradients!")  # :85-88
            # :90 -- eager zero grad allocated with the constructor's `dtype` ARGUMENT: a leaf made
            # without an explicit dtype gets a float64 grad even when its data is float32.
            self.grad = np.zeros(self.value.shape, dtype=dtype)
            _State.counter += 1
            self.idx = _State.counter
            _State.nodes.append(self)
        else:
            self.idx = -1

    # -- metadata
    shape = property(lambda s: s.value.shape)
    ndim = property(lambda s: s.value.ndim)
    dtype = property(lambda s: s.value.dtype)
    size = property(lambda s: s.value.size)

    def item(self):
        return self.value.item()

    def numpy(self):
        return self.value.copy()

    def zero_grad(self):
        self.grad[...] = 0.0   # :380-383

    # -- operators (tensor.py:218-264)
    def __add__(s, o): return add(s, o)
    def __radd__(s, o): return add(o, s)
    def __sub__(s, o): return sub(s, o)
    def __rsub__(s, o): return sub(o, s)
    def __mul__(s, o): return mul(s, o)
    def __rmul__(s, o): return mul(o, s)
    def __truediv__(s, o): return div(s, o)
    def __rtruediv__(s, o): return div(o, s)
    def __pow__(s, o): return pow(s, o)
    def __rpow__(s, o): return pow(o, s)
    def __matmul__(s, o): return matmul(s, o)
    def __neg__(s): return mul(-1, s)         # :255-256
    def __getitem__(s, k): return getitem(s, k)
    def reshape(s, *shape): return reshape(s, shape)
    def transpose(s, *axes): return transpose(s, axes if len(axes) else None)
    def swapaxes(s, a, b): return swapaxes(s, a, b)
    def sum(s, axis=None, keepdims=False): return sum(s, axis, keepdims)
    def mean(s, axis=None, keepdims=False): return mean(s, axis, keepdims)
    def max(s, axis=None, keepdims=False): return max(s, axis, keepdims)
    def min(s, axis=None, keepdims=False): return min(s, axis, keepdims)

    def backward(self, retain_graph=False):
        backward(self, retain_graph)


def _node(value, parents):
    """Result of an op: tracked iff grad is enabled and any parent is (:438-447, :498-508)."""
    req = _State.grad_enabled and any(p.requires_grad for p, _ in parents)
    v = Var.__new__(Var)
    v.value = value
    v.requires_grad = bool(req)
    v.parents = [(p, f) for p, f in parents] if req else []
    v.grad = None
    v.name = None
    if req:
        v.grad = np.zeros(value.shape, dtype=value.dtype)
        _State.counter += 1
        v.idx = _State.counter
        _State.nodes.append(v)
    else:
        v.idx = -1
    return v


def _pair(x, y):
    """Scalar operands take the tensor operand's dtype (:488-493)."""
    if not isinstance(x, Var) and isinstance(y, Var):
        x = Var(x, dtype=y.dtype)
    elif isinstance(x, Var) and not isinstance(y, Var):
        y = Var(y, dtype=x.dtype)
    elif not isinstance(x, Var):
        x, y = Var(x), Var(y)
    return x, y


# ---- binary elementwise (:535-640) ------------------------------------------------------
def add(x, y):
    x, y = _pair(x, y)
    return _node(x.value + y.value, [(x, lambda g: g[...]), (y, lambda g: g[...])])


def sub(x, y):
    x, y = _pair(x, y)
    return _node(x.value - y.value, [(x, lambda g: g[...]), (y, lambda g: -g)])


def mul(x, y):
    x, y = _pair(x, y)
    return _node(x.value * y.value, [(x, lambda g: y.value * g), (y, lambda g: x.value * g)])


def div(x, y):
    x, y = _pair(x, y)
    out = x.value / y.value
    return _node(out, [(x, lambda g: g / y.value), (y, lambda g: -out * (g / y.value))])


def pow(x, y):  # noqa: A001
    x, y = _pair(x, y)
    out = x.value ** y.value
    return _node(out, [(x, lambda g: (out * y.value / x.value) * g),
                       (y, lambda g: out * np.log(x.value) * g)])


def maximum(x, y):
    x, y = _pair(x, y)
    out = np.maximum(x.value, y.value)
    # :814-815 -- (out == input) * g for BOTH edges: relu'(0) = 1
    return _node(out, [(x, lambda g: (out == x.value) * g), (y, lambda g: (out == y.value) * g)])


def minimum(x, y):
    x, y = _pair(x, y)
    out = np.minimum(x.value, y.value)
    # :822-823 compares the array to the Tensor OBJECT -> always False -> zero gradient
    zero = lambda g: np.zeros_like(out) * g  # noqa: E731
    return _node(out, [(x, zero), (y, zero)])


# ---- matmul (:643-676) -------------------------------------------------------------------
def matmul(x, y):
    x, y = _pair(x, y)
    ea, eb = x.ndim < 2, y.ndim < 2

    def fix(g):
        if ea:
            g = np.expand_dims(g, 0)
        if eb:
            g = np.expand_dims(g, -1)
        return g

    def gx(g):
        g = fix(g)
        r = g @ (np.atleast_2d(y.value) if eb else y.value.swapaxes(-1, -2))
        return r[0] if ea else r

    def gy(g):
        g = fix(g)
        r = (np.atleast_2d(x.value) if ea else x.value).swapaxes(-1, -2) @ g
        return r[..., 0] if eb else r

    return _node(x.value @ y.value, [(x, gx), (y, gy)])


# ---- unary (:679-692, 776-832, 996-1019) ---------------------------------------------------
def _un(x):
    return x if isinstance(x, Var) else Var(x)


def exp(x):
    x = _un(x)
    out = np.exp(x.value)
    return _node(out, [(x, lambda g: out * g)])


def log(x):
    x = _un(x)
    return _node(np.log(x.value), [(x, lambda g: g / x.value)])


def abs(x):  # noqa: A001
    x = _un(x)

    def vjp(g):  # :691-692 calls xp.sign on the Tensor object -> TypeError in the reference
        raise TypeError("abs.grad_fn: sign() of a Tensor is undefined in the reference")

    return _node(np.abs(x.value), [(x, vjp)])


def sign(x):
    x = _un(x)
    out = np.sign(x.value)
    return _node(out, [(x, lambda g: np.zeros(out.shape, dtype=out.dtype))])


def sigmoid(x):
    x = _un(x)
    v = x.value
    out = np.zeros(v.shape, dtype=v.dtype)
    out[v > 0] = 1 / (1 + np.exp(-v[v > 0]))
    out[v <= 0] = 1 - 1 / (1 + np.exp(v[v <= 0]))
    return _node(out, [(x, lambda g: out * (1 - out) * g)])


def tanh(x):
    x = _un(x)
    v = x.value
    out = np.zeros(v.shape, dtype=v.dtype)
    out[v > 0] = 2 / (1 + np.exp(-2 * v[v > 0])) - 1
    out[v <= 0] = 1 - 2 / (1 + np.exp(2 * v[v <= 0]))
    return _node(out, [(x, lambda g: (1 - out ** 2) * g)])


def sqrt(x):      # core/function.py:4-6
    return pow(x, 0.5)


def square(x):    # core/function.py:9-11
    return mul(x, x)


# ---- reductions (:695-773) -----------------------------------------------------------------
def _reduce(x, axis, keepdims, fn, vjp_of):
    x = _un(x)
    out = getattr(np, fn)(x.value, axis=axis, keepdims=keepdims)
    out = np.asarray(out)

    def vjp(g):
        if not (axis is None or keepdims):
            g = np.expand_dims(g, axis=axis)
        return vjp_of(x.value, out, g)

    return _node(out, [(x, vjp)])


def sum(x, axis=None, keepdims=False):  # noqa: A001
    return _reduce(x, axis, keepdims, "sum", lambda xv, out, g: np.broadcast_to(g, xv.shape))


def mean(x, axis=None, keepdims=False):
    return _reduce(x, axis, keepdims, "mean",
                   lambda xv, out, g: np.broadcast_to(g, xv.shape) * out.size / xv.size)


def _extreme(x, axis, keepdims, fn):
    x = _un(x)
    out = np.asarray(getattr(np, fn)(x.value, axis=axis, keepdims=keepdims))

    def vjp(g):
        full, gg = out, g
        if not (axis is None or keepdims):
            full = np.expand_dims(out, axis=axis)
            gg = np.expand_dims(g, axis=axis)
        return (full == x.value) * gg   # ties: every maximal position receives g

    return _node(out, [(x, vjp)])


def max(x, axis=None, keepdims=False):  # noqa: A001
    return _extreme(x, axis, keepdims, "max")


def min(x, axis=None, keepdims=False):  # noqa: A001
    return _extreme(x, axis, keepdims, "min")


def argmax(x, axis=None, keepdims=False):
    return Var(np.argmax(_un(x).value, axis=axis, keepdims=keepdims))


def argmin(x, axis=None, keepdims=False):
    return Var(np.argmin(_un(x).value, axis=axis, keepdims=keepdims))


# ---- views (:836-901) and core/function.py helpers ------------------------------------------
def reshape(x, shape):
    x = _un(x)
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
        shape = tuple(shape[0])
    return _node(x.value.reshape(shape), [(x, lambda g: g.reshape(x.shape))])


def transpose(x, axes=None):
    x = _un(x)
    if axes is not None and len(axes) == 1 and isinstance(axes[0], (tuple, list)):
        axes = tuple(axes[0])
    inv = None if axes is None else tuple(np.argsort(axes))
    return _node(x.value.transpose(axes), [(x, lambda g: g.transpose(inv) if inv else g.transpose())])


def swapaxes(x, a, b):
    x = _un(x)
    return _node(x.value.swapaxes(a, b), [(x, lambda g: g.swapaxes(a, b))])


def unsqueeze(x, axis):   # core/function.py:226-236
    return reshape(x, np.expand_dims(np.empty(x.shape, dtype=bool), axis).shape)


def squeeze(x, axis=None):  # core/function.py:239-259
    return reshape(x, np.squeeze(np.empty(x.shape, dtype=bool), axis).shape)


def getitem(x, key):
    """:904-940 -- gradient is scatter-ASSIGN: duplicates keep the last write."""
    x = _un(x)
    if isinstance(key, tuple):
        key = tuple(k.value if isinstance(k, Var) else k for k in key)
    elif isinstance(key, Var):
        key = key.value

    def vjp(g):
        full = np.zeros(x.shape, dtype=x.dtype)
        full[key] = g
        return full

    return _node(x.value[key], [(x, vjp)])


def concat(vars_, axis=0):   # :943-993
    vals = [v.value for v in vars_]
    bounds = np.cumsum([0] + [v.shape[axis] for v in vals])

    def make(i):
        def vjp(g):
            sl = [slice(None)] * g.ndim
            sl[axis] = slice(bounds[i], bounds[i + 1])
            return g[tuple(sl)]
        return vjp

    return _node(np.concatenate(vals, axis=axis), [(v, make(i)) for i, v in enumerate(vars_)])


def split(x, sections, axis=0):   # core/function.py:129-166 (equal division only)
    n = x.shape[axis]
    assert n % sections == 0, "array split does not result in an equal division"
    step = n // sections
    outs = []
    for i in range(sections):
        sl = [slice(None)] * x.ndim
        sl[axis] = slice(i * step, (i + 1) * step)
        outs.append(getitem(x, tuple(sl)))
    return outs


# ---- the engine (:327-375) -------------------------------------------------------------------
def backward(root: Var, retain_graph=False):
    if root not in _State.nodes:
        raise ValueError("Auto-grad is failed because current node is not in graph.")
    if root.size > 1:
        raise ValueError("backward should be called only on a scalar.")
    pos = len(_State.nodes) - 1 - _State.nodes[::-1].index(root)
    root.grad = np.ones(root.shape, dtype=root.dtype)
    # the reference walks EVERY earlier node of the global list, not just ancestors (:353-356)
    for node in _State.nodes[pos::-1]:
        for parent, vjp in node.parents:
            if not parent.requires_grad:
                continue
            g = vjp(node.grad)
            if g.shape != parent.shape:   # un-broadcast, left-indexed quirk preserved (:360-370)
                d1, d2 = g.ndim, parent.ndim
                g = g.sum(axis=tuple(i for i in range(d2) if parent.shape[i] == 1), keepdims=True)
                if d1 != d2:
                    g = g.sum(tuple(range(d1 - d2)))
            parent.grad += g
        if not retain_graph and node.parents:
            node.parents = []
            _State.nodes.remove(node)
