"""Tensor, tape and reverse-mode engine -- the surface of pydynet/core/tensor.py, rebuilt.

What is kept (callers depend on it): `Tensor(data, dtype, copy, device, requires_grad)`, public
`.data/.grad/.last/.requires_grad/.device/.xp`, operator overloads, the operator plug-in
protocol (`forward_` + `grad_fn(input_tensor, upstream_grad)` called once per input edge), the
exceptions (tensor.py:73-75, 85-88, 267-269, 346-351) and the `retain_graph` contract.

What is new (the reference's mechanics are what make it slow):
  * the tape is a weak, O(1) registry and `backward()` walks only the ANCESTORS of the root in
    reverse creation order (the reference visits every earlier node of a global list);
  * op nodes get their gradient lazily (no zero-filled buffer per node); the first
    contribution is adopted without a copy, later ones accumulate;
  * n-ary fused nodes may return all input gradients at once (`backward_all`);
  * leaves fire a `grad-ready` hook when their last contribution lands (used by the
    data-parallel bucketed all-reduce to overlap communication with backward).
Every array expression goes through `self.xp` (numpy on "cpu", `hipnp` -> HIP kernels on GPU).
"""
from __future__ import annotations

import weakref

import numpy as np

from ..autograd import is_grad_enable, no_grad
from ..cuda import Device


class Graph:
    """Registry of grad-tracked tensors (weak: a dropped dead branch disappears by itself)."""
    _nodes: "weakref.WeakValueDictionary[int, Tensor]" = weakref.WeakValueDictionary()
    _counter = 0

    @classmethod
    def _add_node(cls, node):
        cls._counter += 1
        node._gid = cls._counter
        cls._nodes[node._gid] = node

    @classmethod
    def _free_node(cls, node):
        node.last = []
        cls._nodes.pop(node._gid, None)

    @classmethod
    def contains(cls, node) -> bool:
        return cls._nodes.get(getattr(node, "_gid", -1)) is node

    @classmethod
    def node_list(cls):
        return [cls._nodes[k] for k in sorted(cls._nodes.keys())]

    @classmethod
    def size(cls):
        return len(cls._nodes)

    @classmethod
    def clear(cls):
        cls._nodes.clear()


def _is_array(x):
    return isinstance(x, np.ndarray) or type(x).__name__ == "ndarray"


_chain = None          # core/fused/chain.py once the fused package is imported (it imports this module)


class Tensor:
    _gid = -1
    _host_scalar = False
    _grad_owned = True
    _grad_hook = None
    _pending_link = False   # True on the deferred links of core/fused/chain.py
    _rope_link = False      # True on the pending links of a rotary embedding written with plain operators (chain.py: rope_link)
    _mul_hook = False       # True on core/fused/pointwise.py's `silu` (pending activation: `silu(gate) * up` becomes one node)
    _reshape_hook = False   # True where core/fused/chain.py wants to see `reshape`: pending projections, the tail of a rotary embedding
    _causal_mask = False    # True on a Tensor built from exactly the additive causal mask of llm/llama/model.py:199-203

    def __init__(self, data, dtype=None, copy=True, device=None, requires_grad=False) -> None:
        if isinstance(data, Tensor):
            raise ValueError("Tensor assignment with another tensor is forbidden.")
        self.copy = copy
        self.device = device if isinstance(device, Device) else Device(device)
        xp = self.device.xp
        with self.device:
            if xp is np:
                if type(data).__module__.startswith("pydynet_amd"):
                    data = data.get()          # a device array handed to a cpu Tensor
                self.data = np.array(data, dtype=dtype, copy=copy)
            else:
                if (isinstance(data, np.ndarray) and data.ndim == 2 and data.shape[0] == data.shape[1] > 1
                        and data.dtype.kind == "f" and np.isneginf(data[0, -1])):
                    # the additive causal mask, recognised on the host while it still is a NumPy array: the attention
                    # chain built from plain operators then takes the kernels' causal schedule (core/fused/chain.py)
                    n = data.shape[0]
                    self._causal_mask = bool(np.array_equal(data, np.triu(np.full((n, n), -np.inf, data.dtype), k=1)))
                self.data = xp.array(data, dtype=dtype, copy=bool(copy))
        self.requires_grad = is_grad_enable() and requires_grad
        self.last = []
        if self.requires_grad:
            if not np.issubdtype(self.data.dtype, np.floating):
                raise TypeError("Only Tensors of floating point dtype can require gradients!")
            with self.device:
                # reference quirk kept: the grad buffer takes the constructor's `dtype` argument
                self.grad = xp.zeros(self.data.shape, dtype=dtype)
            Graph._add_node(self)
        else:
            self.grad = None

    # ---- metadata ---------------------------------------------------------------------
    @property
    def is_leaf(self) -> bool:
        return not self.requires_grad or len(self.last) == 0

    @property
    def shape(self): return self.data.shape
    @property
    def ndim(self): return self.data.ndim
    @property
    def dtype(self): return self.data.dtype
    @property
    def size(self): return self.data.size
    @property
    def strides(self): return self.data.strides
    @property
    def T(self): return self.transpose()
    @property
    def xp(self): return self.device.xp

    def __len__(self): return len(self.data)

    def __repr__(self) -> str:
        dev = "" if self.device.device == "cpu" else f", device={self.device}"
        return f"Tensor({self.numpy()}, requires_grad={self.requires_grad}{dev})"

    # ---- conversions ------------------------------------------------------------------
    def astype(self, new_type):
        assert not self.requires_grad
        with self.device:
            return Tensor(self.data.astype(new_type), new_type, copy=None, device=self.device)

    def numpy(self):
        return self.data.copy() if isinstance(self.data, np.ndarray) else self.data.get()

    def item(self):
        return self.data.item()

    def to(self, device):
        device = device if isinstance(device, Device) else Device(device)
        if self.device != device:
            def move(a):
                host = a if isinstance(a, np.ndarray) else a.get()
                with device:
                    return host if device.xp is np else device.xp.asarray(host)
            self.data = move(self.data)
            if self.requires_grad and self.grad is not None:
                self.grad = move(self.grad)
            self.device = device
        return self

    def cpu(self): return self.to("cpu")
    def cuda(self, id: int = 0): return self.to(f"cuda:{id}")
    def hip(self, id: int = 0): return self.to(f"hip:{id}")

    def zero_grad(self):
        with self.device:
            self.grad[...] = 0.

    # ---- views / reductions / operators -------------------------------------------------
    def reshape(self, *new_shape):
        # a pending projection regrouped; the rotary embedding: its first reshape (to pairs), its unsqueezes, its tail (chain.py)
        if _chain is not None and (self._reshape_hook or (len(new_shape) == 5 and new_shape[-1] == 2)):
            r = _chain.on_reshape(self, new_shape)
            if r is not None:
                return r
        return reshape(self, new_shape)
    def transpose(self, *axes): return transpose(self, axes if len(axes) != 0 else None)
    def swapaxes(self, axis1, axis2): return swapaxes(self, axis1, axis2)
    def max(self, axis=None, keepdims=False): return max(self, axis, keepdims)
    def min(self, axis=None, keepdims=False): return min(self, axis, keepdims)
    def mean(self, axis=None, keepdims=False): return mean(self, axis, keepdims)
    def sum(self, axis=None, keepdims=False): return sum(self, axis, keepdims)
    def argmax(self, axis=None, keepdims=False): return argmax(self, axis, keepdims)
    def argmin(self, axis=None, keepdims=False): return argmin(self, axis, keepdims)

    def __add__(self, x):
        if _chain is not None and self._pending_link:
            r = _chain.on_add(self, x)
            if r is not None:
                return r
        if self._rope_link and _chain is not None:
            r = _chain.on_addsub(self, x, "add")
            if r is not None:
                return r
        return add(self, x)
    def __radd__(self, x): return add(x, self)
    def __sub__(self, x):
        if self._rope_link and _chain is not None:
            r = _chain.on_addsub(self, x, "sub")
            if r is not None:
                return r
        return sub(self, x)
    def __rsub__(self, x): return sub(x, self)
    def __mul__(self, x):
        if _chain is not None and (self._mul_hook or getattr(x, "_mul_hook", False)):    # silu(gate) * up (chain.py)
            r = _chain.on_mul(self, x)
            if r is not None:
                return r
        return mul(self, x)
    def __rmul__(self, x): return mul(x, self)
    def __matmul__(self, x):
        # (the reference's attention chain built from plain operators becomes one fused node: core/fused/chain.py)
        if _chain is not None and (self._pending_link or type(self) is transpose) and isinstance(x, Tensor):
            r = _chain.on_matmul(self, x)
            if r is not None:
                return r
        return matmul(self, x)
    def __rmatmul__(self, x): return matmul(x, self)
    def __truediv__(self, x):
        if _chain is not None and self._pending_link:
            r = _chain.on_div(self, x)
            if r is not None:
                return r
        return div(self, x)
    def __rtruediv__(self, x): return div(x, self)
    def __pow__(self, x): return pow(self, x)
    def __rpow__(self, x): return pow(x, self)
    def __pos__(self): return 1 * self
    def __neg__(self): return -1 * self
    def __abs__(self): return abs(self)
    def __getitem__(self, key):
        if self._rope_link and _chain is not None:
            r = _chain.on_getitem(self, key)
            if r is not None:
                return r
        return _get_slice(self, key)

    def _inplace(self, *others, func):
        if self.requires_grad and is_grad_enable():
            raise ValueError("In-place operation is forbidden in node requires grad.")
        others = tuple(o.data if isinstance(o, Tensor) else o for o in others)
        with self.device:
            r = func(*others)
        if r is not None and r is not NotImplemented and _is_array(r):
            self.data = r
        return self

    def __setitem__(self, key, value):
        if isinstance(key, tuple):
            key = tuple(k.data if isinstance(k, Tensor) else k for k in key)
        elif isinstance(key, Tensor):
            key = key.data
        if self.requires_grad and is_grad_enable():
            raise ValueError("In-place operation is forbidden in node requires grad.")
        with self.device:
            self.data[key] = value.data if isinstance(value, Tensor) else value

    def __iadd__(self, other): return self._inplace(other, func=self.data.__iadd__)
    def __isub__(self, other): return self._inplace(other, func=self.data.__isub__)
    def __imul__(self, other): return self._inplace(other, func=self.data.__imul__)
    def __itruediv__(self, other): return self._inplace(other, func=self.data.__itruediv__)
    def __imatmul__(self, other): return self._inplace(other, func=self.data.__imatmul__)

    def _compare(self, other, func):
        with self.device, no_grad():
            o = other.data if isinstance(other, Tensor) else other
            return Tensor(func(self.data, o), np.bool_, None, self.device, False)

    def eq(self, other): return self._compare(other, lambda x, y: x == y)
    def ne(self, other): return self._compare(other, lambda x, y: x != y)
    def __lt__(self, other): return self._compare(other, lambda x, y: x < y)
    def __le__(self, other): return self._compare(other, lambda x, y: x <= y)
    def __gt__(self, other): return self._compare(other, lambda x, y: x > y)
    def __ge__(self, other): return self._compare(other, lambda x, y: x >= y)

    def _build_edge(self, node):
        node.last.append(self)

    # ---- reverse mode -------------------------------------------------------------------
    def backward(self, retain_graph: bool = False):
        if not Graph.contains(self):
            raise ValueError("Auto-grad is failed because current node is not in graph.")
        if self.size > 1:
            raise ValueError("backward should be called only on a scalar.")
        xp = self.xp
        with self.device:
            self.grad = xp.ones(self.shape, dtype=self.dtype)
            self._grad_owned = True
            order, pending = _ancestors(self)
            for node in order:
                g = node.grad
                inputs = node.last
                if not inputs:
                    continue
                if g is not None:
                    folded = ()
                    if hasattr(node, "backward_all"):
                        if node.folds_existing:
                            # hand the node the gradients its non-leaf inputs already hold: it may
                            # add them inside its own kernel (GEMM / norm epilogue) instead of the
                            # engine running a separate accumulation pass afterwards
                            node._existing = [inp.grad if (inp.requires_grad and inp.last) else None
                                              for inp in inputs]
                            folded = node._folded = set()
                        grads = node.backward_all(g)
                        if node.folds_existing:
                            node._existing = None
                    else:
                        grads = [node.grad_fn(inp, g) if inp.requires_grad else None for inp in inputs]
                    for i, (inp, add_grad) in enumerate(zip(inputs, grads)):
                        if inp.requires_grad and add_grad is not None:
                            if i in folded:
                                inp.grad, inp._grad_owned = add_grad, True
                            else:
                                _accumulate(inp, add_grad, xp)
                for inp in inputs:
                    if inp.requires_grad and not inp.last:       # a leaf: count down its edges
                        left = pending.get(inp._gid, 0) - 1
                        pending[inp._gid] = left
                        if left == 0 and inp._grad_hook is not None:
                            inp._grad_hook(inp)
                if not retain_graph:
                    Graph._free_node(node)
                    if node is not self:
                        node.grad = None


def _ancestors(root):
    """Grad-tracked ancestors of `root` in reverse creation order (a valid reverse topological
    order: inputs are always created before their consumers) + edge counts of the leaves."""
    seen = {root._gid: root}
    pending = {}
    stack = [root]
    while stack:
        n = stack.pop()
        for inp in n.last:
            if not inp.requires_grad:
                continue
            if not inp.last:
                pending[inp._gid] = pending.get(inp._gid, 0) + 1
            if inp._gid not in seen:
                seen[inp._gid] = inp
                stack.append(inp)
    return [seen[k] for k in sorted(seen, reverse=True)], pending


def _unbroadcast(g, shape):
    if g.shape == tuple(shape):
        return g
    lead = g.ndim - len(shape)
    if lead > 0:
        g = g.sum(axis=tuple(range(lead)))
    axes = tuple(i for i, s in enumerate(shape) if s == 1 and g.shape[i] != 1)
    if axes:
        g = g.sum(axis=axes, keepdims=True)
    return g


def _accumulate(t, add_grad, xp):
    fresh = add_grad.shape != tuple(t.shape)
    if fresh:
        add_grad = _unbroadcast(add_grad, t.shape)
    if not t.last and t.grad is not None and t._grad_owned:
        # a true leaf accumulates in place into its own buffer.  (A stale op node that is a leaf of THIS graph
        # adopted its first contribution below and may alias another node's gradient: `_grad_owned` is False
        # then and the sum is formed out of place.)
        t.grad += add_grad
    elif t.grad is None:
        # first contribution to an op node: adopt, no copy.  (Also an op node whose own graph was
        # already walked and freed -- e.g. a hidden state carried into the next batch: it is a leaf of
        # the new graph, the gradient stops here and stays readable, as in the reference.)
        t.grad = add_grad
        t._grad_owned = fresh
    elif t._grad_owned:
        t.grad += add_grad
        if getattr(t.grad, "_aux", None) is not None:
            t.grad._aux = None           # a producer's note about the array (column sums, relu bits applied) no longer holds
    else:                                # adopted array may alias another node's grad
        t.grad = t.grad + add_grad
        t._grad_owned = True


# ---------------------------------------------------------------------------------------
# operator protocol
# ---------------------------------------------------------------------------------------
def _as_operand(v, like: Tensor):
    """Python / NumPy scalars become host-scalar operands carrying the tensor operand's dtype
    (tensor.py:488-493) -- kept on the host so no device transfer is paid per scalar."""
    if like.device.is_hip and np.ndim(v) == 0:
        t = Tensor.__new__(Tensor)
        t.data = np.array(v, dtype=like.dtype)
        t.device, t.requires_grad, t.grad, t.last, t.copy = like.device, False, None, [], None
        t._host_scalar = True
        return t
    return Tensor(v, dtype=like.dtype, device=like.device)


class _Operator(Tensor):
    """n-ary differentiable node.  Subclasses implement `forward_(*inputs) -> array` and either
    `grad_fn(input, grad) -> array` (per edge) or `backward_all(grad) -> [array|None]`.  A node with
    `folds_existing = True` receives `self._existing` (gradients its inputs already hold) during
    backward_all and lists in `self._folded` the inputs whose returned gradient includes them."""

    folds_existing = False

    def _init_node(self, data, device, inputs):
        self.data = data
        self.device = device
        self.copy = None
        self.grad = None
        track = False
        if is_grad_enable():
            for t in inputs:
                if t.requires_grad:
                    track = True
                    break
        self.requires_grad = track
        if track:
            if not np.issubdtype(data.dtype, np.floating):
                raise TypeError("Only Tensors of floating point dtype can require gradients!")
            self.last = list(inputs)
            Graph._add_node(self)
        else:
            self.last = []

    def __init__(self, *inputs):
        self.device = inputs[0].device
        with self.device:
            data = self.forward_(*inputs)
        if not _is_array(data):          # NumPy returns scalars for 0-d results
            data = np.asarray(data)
        self._init_node(data, self.device, inputs)

    def forward_(self, *inputs):
        raise NotImplementedError

    def grad_fn(self, x, grad):
        raise NotImplementedError

    def __repr__(self) -> str:
        return f"Tensor({self.numpy()}, op={self.__class__.__name__})"


class _UnaryOperator(_Operator):
    def __init__(self, x) -> None:
        if not isinstance(x, Tensor):
            x = Tensor(x)
        super().__init__(x)

    def forward(self, x):
        with self.device:
            return self.forward_(x)


class _BinaryOperator(_Operator):
    def __init__(self, x, y) -> None:
        xt, yt = isinstance(x, Tensor), isinstance(y, Tensor)
        if not xt and yt:
            x = _as_operand(x, y)
        elif xt and not yt:
            y = _as_operand(y, x)
        elif not xt:
            x, y = Tensor(x), Tensor(y)
        assert x.device == y.device
        super().__init__(x, y)

    def forward(self, x, y):
        with self.device:
            return self.forward_(x, y)


def _binary(name, fwd, grad_first, grad_second, doc):
    def forward_(self, x, y):
        return fwd(self.xp, x.data, y.data)

    def grad_fn(self, node, grad):
        return grad_first(self, grad) if node is self.last[0] else grad_second(self, grad)

    return type(name, (_BinaryOperator,), {"forward_": forward_, "grad_fn": grad_fn, "__doc__": doc})


add = _binary("add", lambda xp, a, b: a + b, lambda s, g: g, lambda s, g: g,
              "x + y  (tensor.py:535-551)")
sub = _binary("sub", lambda xp, a, b: a - b, lambda s, g: g, lambda s, g: -g,
              "x - y  (tensor.py:554-569)")
mul = _binary("mul", lambda xp, a, b: a * b, lambda s, g: s.last[1].data * g,
              lambda s, g: s.last[0].data * g, "x * y  (tensor.py:572-596)")
div = _binary("div", lambda xp, a, b: a / b, lambda s, g: g / s.last[1].data,
              lambda s, g: -s.data * (g / s.last[1].data), "x / y  (tensor.py:599-618)")
pow = _binary("pow", lambda xp, a, b: a ** b,  # noqa: A001
              lambda s, g: (s.data * s.last[1].data / s.last[0].data) * g,
              lambda s, g: s.data * s.xp.log(s.last[0].data) * g, "x ** y  (tensor.py:621-640)")
# (out == input) * g on both edges: ties / relu at 0 pass the gradient (tensor.py:808-815)
maximum = _binary("maximum", lambda xp, a, b: xp.maximum(a, b),
                  lambda s, g: (s.data == s.last[0].data) * g,
                  lambda s, g: (s.data == s.last[1].data) * g, "elementwise maximum")
# reference quirk: minimum's grad_fn compares against the Tensor object -> gradient is 0
# everywhere (tensor.py:822-823); kept for parity.
minimum = _binary("minimum", lambda xp, a, b: xp.minimum(a, b),
                  lambda s, g: 0.0 * g, lambda s, g: 0.0 * g, "elementwise minimum (zero gradient)")


class matmul(_BinaryOperator):
    """NumPy-rule matmul; dA = g @ B^T, dB = A^T @ g with the 1-D fix-ups (tensor.py:643-676)."""

    def forward_(self, x, y):
        self.expand_a, self.expand_b = x.ndim < 2, y.ndim < 2
        return x.data @ y.data

    def grad_fn(self, node, grad):
        xp = self.xp
        if self.expand_a:
            grad = xp.expand_dims(grad, 0)
        if self.expand_b:
            grad = xp.expand_dims(grad, -1)
        a, b = self.last[0].data, self.last[1].data
        if node is self.last[0]:
            r = grad @ (xp.atleast_2d(b) if self.expand_b else b.swapaxes(-1, -2))
            return r[0] if self.expand_a else r
        r = (xp.atleast_2d(a) if self.expand_a else a).swapaxes(-1, -2) @ grad
        return r[..., 0] if self.expand_b else r


def _unary(name, fwd, grad, doc):
    return type(name, (_UnaryOperator,), {
        "forward_": lambda self, x: fwd(self.xp, x.data),
        "grad_fn": lambda self, x, g: grad(self, x, g), "__doc__": doc})


exp = _unary("exp", lambda xp, a: xp.exp(a), lambda s, x, g: s.data * g, "tensor.py:776-789")
log = _unary("log", lambda xp, a: xp.log(a), lambda s, x, g: g / x.data, "tensor.py:792-805")
sign = _unary("sign", lambda xp, a: xp.sign(a),
              lambda s, x, g: s.xp.zeros(s.shape, dtype=s.dtype), "tensor.py:826-832")


def _abs_grad(s, x, g):
    # the reference calls xp.sign on the Tensor object (tensor.py:691-692) and raises; same here
    raise TypeError("abs.grad_fn is undefined in the reference (sign of a Tensor object)")


abs = _unary("abs", lambda xp, a: xp.abs(a), _abs_grad, "tensor.py:679-692")  # noqa: A001


def _piecewise(v, pos, neg):
    out = np.zeros(v.shape, dtype=v.dtype)
    m = v > 0
    out[m] = pos(v[m])
    out[~m] = neg(v[~m])
    return out


class sigmoid(_UnaryOperator):
    """Overflow-safe piecewise sigmoid (tensor.py:996-1006); one fused kernel on HIP."""

    def forward_(self, x):
        if self.xp is np:
            return _piecewise(x.data, lambda t: 1 / (1 + np.exp(-t)), lambda t: 1 - 1 / (1 + np.exp(t)))
        return self.xp.sigmoid(x.data)

    def grad_fn(self, x, grad):
        return self.data * (1 - self.data) * grad


class tanh(_UnaryOperator):
    """Overflow-safe piecewise tanh (tensor.py:1009-1019)."""

    def forward_(self, x):
        if self.xp is np:
            return _piecewise(x.data, lambda t: 2 / (1 + np.exp(-2 * t)) - 1, lambda t: 1 - 2 / (1 + np.exp(2 * t)))
        return self.xp.tanh(x.data)

    def grad_fn(self, x, grad):
        return (1 - self.data ** 2) * grad


# ---- reductions (tensor.py:695-773) ------------------------------------------------------
class _ReduceOperator(_UnaryOperator):
    _func = None

    def __init__(self, x, axis=None, keepdims=False, func: str = None):
        self.axis, self.keepdims = axis, keepdims
        self._scalar_or_keeps = axis is None or keepdims
        self._fname = func or self._func
        super().__init__(x)

    def forward_(self, x):
        return getattr(self.xp, self._fname)(x.data, axis=self.axis, keepdims=self.keepdims)

    def _full(self, arr):
        return arr if self._scalar_or_keeps else self.xp.expand_dims(arr, axis=self.axis)


class sum(_ReduceOperator):  # noqa: A001
    _func = "sum"

    def grad_fn(self, x, grad):
        return self.xp.broadcast_to(self._full(grad), x.shape)


class mean(_ReduceOperator):
    _func = "mean"

    def grad_fn(self, x, grad):
        return self.xp.broadcast_to(self._full(grad), x.shape) * (self.size / x.size)


class max(_ReduceOperator):  # noqa: A001
    _func = "max"

    def grad_fn(self, x, grad):       # every tied position receives the gradient
        return (self._full(self.data) == x.data) * self._full(grad)


class min(_ReduceOperator):  # noqa: A001
    _func = "min"

    def grad_fn(self, x, grad):
        return (self._full(self.data) == x.data) * self._full(grad)


class argmax(_ReduceOperator):
    _func = "argmax"


class argmin(_ReduceOperator):
    _func = "argmin"


# ---- views (tensor.py:836-901) -------------------------------------------------------------
class reshape(_UnaryOperator):
    def __init__(self, x, new_shape) -> None:
        if len(new_shape) == 1 and isinstance(new_shape[0], (tuple, list)):
            new_shape = tuple(new_shape[0])
        self.new_shape = new_shape
        self._src = x if isinstance(x, Tensor) else None      # (core/fused/chain.py looks through the view, also without a tape)
        super().__init__(x)

    def forward_(self, x): return x.data.reshape(self.new_shape)
    def grad_fn(self, x, grad): return grad.reshape(x.shape)


class transpose(_UnaryOperator):
    def __init__(self, x, axes=None) -> None:
        if axes is not None and len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        self.axes = axes
        self._src = x if isinstance(x, Tensor) else None      # (core/fused/chain.py looks through the view)
        super().__init__(x)

    def forward_(self, x): return x.data.transpose(self.axes)

    def grad_fn(self, x, grad):
        if self.axes is None:
            return grad.transpose()
        return grad.transpose(tuple(int(i) for i in np.argsort(self.axes)))


class swapaxes(_UnaryOperator):
    def __init__(self, x, axis1, axis2) -> None:
        self.axis1, self.axis2 = axis1, axis2
        super().__init__(x)

    def forward_(self, x): return x.data.swapaxes(self.axis1, self.axis2)
    def grad_fn(self, x, grad): return grad.swapaxes(self.axis1, self.axis2)


class _get_slice(_UnaryOperator):
    """Indexing; the gradient is a scatter-ASSIGN into zeros (tensor.py:904-940): with duplicate
    indices the last write wins -- reproduced bit-exactly by the HIP scatter kernel."""

    def __init__(self, x, key) -> None:
        if isinstance(key, tuple):
            key = tuple(k.data if isinstance(k, Tensor) else k for k in key)
        elif isinstance(key, Tensor):
            key = key.data
        self.key = key
        super().__init__(x)

    def forward_(self, x): return x.data[self.key]

    def grad_fn(self, x, grad):
        full = self.xp.zeros(x.shape, dtype=x.dtype)
        full[self.key] = grad
        return full


class concat(_Operator):
    """xp.concatenate; each input receives its slice of the gradient (tensor.py:943-993)."""

    _reshape_hook = True

    def __new__(cls, tensors=(), axis=0):
        # two pending halves of a rotary embedding (chain.py: on_concat); anything else -- other sequences, iterators that
        # __init__ must still be able to walk -- goes straight on
        if cls is concat and _chain is not None and isinstance(tensors, (list, tuple)) and len(tensors) == 2 and \
                getattr(tensors[0], "_rope_link", False):
            r = _chain.on_concat(tensors, axis)
            if r is not None:
                return r
        return object.__new__(cls)

    def __init__(self, tensors, axis=0) -> None:
        tensors = list(tensors)
        for t in tensors:
            assert isinstance(t, Tensor), "Concatenate elements in 'tensors' must be 'Tensor'"
            assert t.device == tensors[0].device
        self.tensors, self.axis = tensors, axis
        self.indices = [0]
        for t in tensors:
            self.indices.append(self.indices[-1] + t.shape[axis])
        super().__init__(*tensors)

    def forward_(self, *tensors):
        return self.xp.concatenate([t.data for t in tensors], axis=self.axis)

    def backward_all(self, grad):
        out = []
        for i in range(len(self.tensors)):
            sl = [slice(None)] * grad.ndim
            sl[self.axis] = slice(self.indices[i], self.indices[i + 1])
            out.append(grad[tuple(sl)])
        return out

    def grad_fn(self, x, grad):
        return self.backward_all(grad)[[t is x for t in self.tensors].index(True)]
