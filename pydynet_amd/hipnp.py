"""``hipnp`` -- the array module a HIP :class:`~pydynet_amd.cuda.Device` returns as ``xp``.

In the reference ``Device.xp`` is ``numpy`` or ``cupy`` (pydynet/cuda.py:89-91) and every
operator is a one-line array expression on ``Tensor.data``.  This module is the MI355X
counterpart of that seam: an ``ndarray`` living in HBM whose operators, reductions,
indexing and assignment all lower to the hand-written HIP kernels of ``libpdnhip.so``
through the C ABI (``include/pdn_hip.h``).  Nothing here computes on the host, and nothing
here needs PyTorch: device memory comes from the library's caching allocator (pdn_malloc),
copies from pdn_memcpy_*, ordering from the per-device compute stream (pdn_compute_stream).

Supported dtypes on the device: float32 (the hot path), float64, int64, int32, bool, and float16 as
a storage type (elementwise math rounds like NumPy's; reductions and matmul compute in float32).
"""
from __future__ import annotations

import builtins as _bi
import ctypes
import math
import numbers

import numpy as np

from . import _lib

# numpy names re-exported so `xp.float32`, `xp.issubdtype(...)` keep working
float16, float32, float64, int64, int32, bool_, floating = (np.float16, np.float32, np.float64, np.int64, np.int32,
                                                            np.bool_, np.floating)
issubdtype = np.issubdtype
newaxis = None
pi, inf = np.pi, np.inf

_DT = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int64): 2,
       np.dtype(np.bool_): 3, np.dtype(np.int32): 4, np.dtype(np.float16): 5}
_BOP = dict(add=0, sub=1, mul=2, div=3, pow=4, maximum=5, minimum=6,
            eq=16, ne=17, lt=18, le=19, gt=20, ge=21)
_UOP = dict(copy=0, neg=1, exp=2, log=3, abs=4, sign=5, sqrt=6, square=7, recip=8,
            sigmoid=9, tanh=10)
_ROP = dict(sum=0, mean=1, max=2, min=3, argmax=4, argmin=5)

_state = {"device": 0, "stream": 0, "streams": {}}


class _Buffer:
    """One block of HBM from the library's caching allocator (pdn_malloc); returned to the cache
    when the last array viewing it is dropped."""

    __slots__ = ("ptr", "nbytes", "device", "__weakref__")

    def __init__(self, nbytes: int):
        p = ctypes.c_void_p()
        _lib.lib().call("pdn_malloc", ctypes.byref(p), int(nbytes))
        self.ptr = p.value or 0
        self.nbytes = int(nbytes)
        self.device = _state["device"]

    def __del__(self):
        ptr, self.ptr = self.ptr, 0
        if ptr:
            try:
                _lib.lib().free(ptr)
            except Exception:                      # interpreter shutdown
                pass


def _dev():
    return f"hip:{_state['device']}"


def set_device(index: int):
    """Make GPU `index` current: allocations, copies and launches go to it and to its compute stream."""
    index = int(index)
    L = _lib.lib()
    L.call("pdn_set_device", index)
    _state["device"] = index
    st = _state["streams"].get(index)
    if st is None:
        h = ctypes.c_void_p()
        L.call("pdn_compute_stream", ctypes.byref(h))
        st = _state["streams"][index] = h.value or 0
    _state["stream"] = st


def current_device() -> int:
    return _state["device"]


def set_stream(handle: int):
    """hipStream_t (as int) every kernel is enqueued on.  The allocator's reuse is ordered on the
    device's compute stream: route work to another stream only for buffers that outlive it."""
    _state["stream"] = int(handle)


def stream() -> int:
    if not _state["streams"]:
        set_device(_state["device"])
    return _state["stream"]


def synchronize():
    _lib.lib().call("pdn_stream_synchronize", stream())


def memory_stats(device=None):
    """Allocator counters of a device: bytes in use / reserved / peak, driver allocations, requests, hits."""
    vals = [ctypes.c_int64() for _ in range(6)]
    _lib.lib().call("pdn_mem_stats", _state["device"] if device is None else int(device),
                    *[ctypes.byref(v) for v in vals])
    keys = ("in_use", "reserved", "peak_in_use", "device_allocs", "requests", "cache_hits")
    return {k: v.value for k, v in zip(keys, vals)}


def empty_cache():
    _lib.lib().call("pdn_empty_cache")


def _dtcode(dt):
    try:
        return _DT[np.dtype(dt)]
    except KeyError:
        raise TypeError(f"HIP backend does not support dtype {np.dtype(dt)} "
                        "(float16, float32, float64, int32, int64, bool only)") from None


def _i64(seq):
    n = len(seq)
    return (ctypes.c_int64 * n)(*seq) if n else (ctypes.c_int64 * 1)()


def _contig_strides(shape):
    st, acc = [], 1
    for s in reversed(shape):
        st.append(acc)
        acc *= s
    return tuple(reversed(st))


# ---------------------------------------------------------------------------------------
# workspace: one growing scratch buffer per process (kernels never allocate)
# ---------------------------------------------------------------------------------------
_ws = {}          # (device, stream) -> _Buffer


def workspace(nbytes: int):
    nbytes = int(nbytes)
    if nbytes <= 0:
        return 0, 0
    key = (_state["device"], _state["stream"])
    g = capturing()
    if g is not None:
        # a step being captured gets scratch of its OWN (from the graph's private pool, alive as long as the
        # graph): the process-wide buffer below may be replaced by a bigger one by any later eager op, and the
        # old block -- whose address the replayed launches still write to -- would be handed to another tensor
        buf = g._ws.get(key)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None and not g.warming:
                g._pinned.append(buf)            # launches captured so far still point into it
            g._ws[key] = buf = _Buffer(_bi.max(nbytes, 1 << 20))
        return buf.ptr, buf.nbytes
    buf = _ws.get(key)
    if buf is None or buf.nbytes < nbytes:
        _ws[key] = buf = _Buffer(_bi.max(nbytes, 1 << 20))
    return buf.ptr, buf.nbytes


_err = {}         # device -> int32[1] array: set by gathers that saw an out-of-range index


def _err_array():
    a = _err.get(_state["device"])
    if a is None:
        a = _err[_state["device"]] = zeros((1,), np.int32)
    return a


def err_flag_ptr() -> int:
    return _err_array()._ptr


def check_index_errors():
    """Raise IndexError if any gather since the last check saw an out-of-range index."""
    a = _err.get(_state["device"])
    if a is not None and int(a.get()[0]) != 0:
        a.fill(0)
        raise IndexError("index out of range in a device gather")


# ---------------------------------------------------------------------------------------
class ndarray:
    """Strided N-d array in HBM.  `_buf` (a `_Buffer`) owns the storage; views share it."""

    __slots__ = ("_buf", "_ptr", "shape", "_strides", "dtype", "_aux", "__weakref__")
    __array_priority__ = 1000.0
    __array_ufunc__ = None     # numpy scalars/arrays defer to our reflected operators

    def __init__(self, buf, ptr, shape, strides, dtype):
        self._buf = buf
        self._ptr = ptr
        self.shape = tuple(int(s) for s in shape)
        self._strides = tuple(int(s) for s in strides)
        self.dtype = np.dtype(dtype)

    # ---- metadata -------------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(math.prod(self.shape))

    @property
    def itemsize(self):
        return self.dtype.itemsize

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def strides(self):
        return tuple(s * self.dtype.itemsize for s in self._strides)

    @property
    def T(self):
        return self.transpose()

    @property
    def data_ptr(self):
        return self._ptr

    @property
    def device_index(self):
        return self._buf.device

    def is_contiguous(self):
        exp = 1
        for s, st in zip(reversed(self.shape), reversed(self._strides)):
            if s != 1 and st != exp:
                return False
            exp *= s
        return True

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    def __repr__(self):
        return f"hipnp.ndarray({self.get()!r}, device=hip:{self.device_index})"

    # ---- host transfer --------------------------------------------------------------
    def get(self) -> np.ndarray:
        """Device -> host copy (synchronises); the counterpart of cupy's `.get()`."""
        # on the stream of the device that OWNS the buffer (its producing kernels were enqueued there), not on
        # whatever device happens to be current: `Device.__exit__` restores the previous GPU after every op
        prev = _state["device"]
        own = self._buf.device if self._buf is not None else prev
        if own != prev:
            set_device(own)
        try:
            a = self if self.is_contiguous() else self.copy()
            host = np.empty(a.shape, dtype=self.dtype)
            if host.size:
                _lib.lib().call("pdn_memcpy_d2h", host.ctypes.data, a._ptr, host.nbytes, stream())
        finally:
            if own != prev:
                set_device(prev)
        return host

    def item(self):
        if self.size != 1:
            raise ValueError("can only convert an array of size 1 to a Python scalar")
        return self.get().reshape(()).item()

    def tolist(self):
        return self.get().tolist()

    def __float__(self):
        return float(self.item())

    def __int__(self):
        return int(self.item())

    def __bool__(self):
        if self.size != 1:
            raise ValueError("The truth value of an array with more than one element is ambiguous.")
        return bool(self.item())

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a.astype(dtype) if dtype is not None else a

    # ---- copies / casts ---------------------------------------------------------------
    def copy(self):
        out = empty(self.shape, self.dtype)
        _cast_into(self, out)
        return out

    def astype(self, dtype, copy=True):
        dtype = np.dtype(dtype)
        if dtype == self.dtype and not copy:
            return self
        out = empty(self.shape, dtype)
        _cast_into(self, out)
        return out

    def fill(self, value):
        _fill(self, value)

    # ---- views ------------------------------------------------------------------------
    def _view(self, shape, strides, elem_off=0):
        return ndarray(self._buf, self._ptr + elem_off * self.dtype.itemsize, shape, strides, self.dtype)

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = _resolve_shape(shape, self.size)
        st = _reshape_strides(self.shape, self._strides, shape)
        if st is None:
            c = self.copy()
            return c._view(shape, _contig_strides(shape))
        out = self._view(shape, st)
        aux = getattr(self, "_aux", None)
        if aux is not None and shape and self.shape and shape[-1] == self.shape[-1]:
            out._aux = aux          # per-column side data (e.g. column sums) survives a row regrouping
        return out

    def transpose(self, *axes):
        if len(axes) == 1 and (axes[0] is None or isinstance(axes[0], (tuple, list))):
            axes = axes[0]
        if axes is None or len(axes) == 0:
            axes = tuple(reversed(range(self.ndim)))
        axes = tuple(a + self.ndim if a < 0 else a for a in axes)
        if sorted(axes) != list(range(self.ndim)):
            raise ValueError("axes don't match array")
        return self._view(tuple(self.shape[a] for a in axes), tuple(self._strides[a] for a in axes))

    def swapaxes(self, a, b):
        ax = list(range(self.ndim))
        ax[a], ax[b] = ax[b], ax[a]
        return self.transpose(ax)

    def squeeze(self, axis=None):
        return self.reshape(np.empty(self.shape, dtype=np.bool_).squeeze(axis).shape)

    def ravel(self):
        return self.reshape(-1)

    flatten = ravel

    # ---- indexing ---------------------------------------------------------------------
    def __getitem__(self, key):
        basic = _basic_index(self, key)
        if basic is not None:
            return basic
        return _advanced_get(self, key)

    def __setitem__(self, key, value):
        if isinstance(key, ndarray) and key.dtype == np.bool_ or (
                isinstance(key, np.ndarray) and key.dtype == np.bool_):
            mask = asarray(key)
            if not isinstance(value, numbers.Number) and not (hasattr(value, "size") and value.size == 1):
                raise NotImplementedError("HIP backend: boolean-mask assignment needs a scalar value")
            v = float(value.item() if hasattr(value, "item") else value)
            shape, (sm, so) = _broadcast([mask, self], out_shape=self.shape)
            _lib.lib().call("pdn_masked_fill", _dtcode(self.dtype), v, len(shape), _i64(shape),
                            mask._ptr, _i64(sm), self._ptr, _i64(so), stream())
            return
        view = _basic_index(self, key)
        if view is None:
            _advanced_set(self, key, value)
            return
        if isinstance(value, (numbers.Number, np.generic, bool)):
            _fill(view, value)
        else:
            _cast_into(asarray(value), view)

    # ---- arithmetic ---------------------------------------------------------------------
    def __add__(self, o): return _binary("add", self, o)
    def __radd__(self, o): return _binary("add", o, self)
    def __sub__(self, o): return _binary("sub", self, o)
    def __rsub__(self, o): return _binary("sub", o, self)
    def __mul__(self, o): return _binary("mul", self, o)
    def __rmul__(self, o): return _binary("mul", o, self)
    def __truediv__(self, o): return _binary("div", self, o)
    def __rtruediv__(self, o): return _binary("div", o, self)
    def __pow__(self, o): return _binary("pow", self, o)
    def __rpow__(self, o): return _binary("pow", o, self)
    def __neg__(self): return _unary("neg", self)
    def __pos__(self): return self.copy()
    def __abs__(self): return _unary("abs", self)
    def __matmul__(self, o): return matmul(self, o)
    def __rmatmul__(self, o): return matmul(asarray(o), self)

    def __iadd__(self, o): return _binary("add", self, o, out=self)
    def __isub__(self, o): return _binary("sub", self, o, out=self)
    def __imul__(self, o): return _binary("mul", self, o, out=self)
    def __itruediv__(self, o): return _binary("div", self, o, out=self)

    def __imatmul__(self, o):
        r = matmul(self, o)
        _cast_into(r, self)
        return self

    def __eq__(self, o): return _binary("eq", self, o)
    def __ne__(self, o): return _binary("ne", self, o)
    def __lt__(self, o): return _binary("lt", self, o)
    def __le__(self, o): return _binary("le", self, o)
    def __gt__(self, o): return _binary("gt", self, o)
    def __ge__(self, o): return _binary("ge", self, o)
    __hash__ = None

    # ---- reductions ---------------------------------------------------------------------
    def sum(self, axis=None, keepdims=False, dtype=None): return _reduce("sum", self, axis, keepdims)
    def mean(self, axis=None, keepdims=False): return _reduce("mean", self, axis, keepdims)
    def max(self, axis=None, keepdims=False): return _reduce("max", self, axis, keepdims)
    def min(self, axis=None, keepdims=False): return _reduce("min", self, axis, keepdims)
    def argmax(self, axis=None, keepdims=False): return _reduce("argmax", self, axis, keepdims)
    def argmin(self, axis=None, keepdims=False): return _reduce("argmin", self, axis, keepdims)


# ---------------------------------------------------------------------------------------
# construction
# ---------------------------------------------------------------------------------------
def empty(shape, dtype=np.float32):
    if isinstance(shape, numbers.Integral):
        shape = (int(shape),)
    shape = tuple(int(s) for s in shape)
    dtype = np.dtype(dtype if dtype is not None else np.float64)
    _dtcode(dtype)
    n = int(math.prod(shape))
    buf = _Buffer(_bi.max(n, 1) * dtype.itemsize)
    return ndarray(buf, buf.ptr, shape, _contig_strides(shape), dtype)


def zeros(shape, dtype=np.float64):
    a = empty(shape, dtype if dtype is not None else np.float64)
    _fill(a, 0)
    return a


def ones(shape, dtype=np.float64):
    a = empty(shape, dtype if dtype is not None else np.float64)
    _fill(a, 1)
    return a


def full(shape, value, dtype=None):
    a = empty(shape, dtype if dtype is not None else np.asarray(value).dtype)
    _fill(a, value)
    return a


def zeros_like(a, dtype=None): return zeros(a.shape, dtype or a.dtype)
def ones_like(a, dtype=None): return ones(a.shape, dtype or a.dtype)
def stacked_view(arrays):
    """A (n, *shape) view over `arrays` when they are contiguous, alike and equally spaced in memory
    (e.g. consecutive parameters of a flat buffer), else None.  Lets n GEMMs that share an operand
    run as one batched launch without copying anything."""
    a0 = arrays[0]
    if not all(isinstance(a, ndarray) and a.shape == a0.shape and a.dtype == a0.dtype and a.is_contiguous()
               for a in arrays):
        return None
    if len(arrays) == 1:
        return a0.reshape((1,) + a0.shape)
    step = arrays[1]._ptr - a0._ptr
    if step % 16 or step == 0 or any(arrays[i + 1]._ptr - arrays[i]._ptr != step for i in range(len(arrays) - 1)):
        return None
    out = ndarray(a0._buf, a0._ptr, (len(arrays),) + a0.shape, (step // a0.dtype.itemsize,) + a0._strides, a0.dtype)
    return out


def empty_like(a, dtype=None): return empty(a.shape, dtype or a.dtype)


def from_numpy(a: np.ndarray) -> ndarray:
    a = np.asarray(a, order="C")          # (np.ascontiguousarray would turn 0-d into 1-d)
    _dtcode(a.dtype)
    buf = _Buffer(_bi.max(a.size, 1) * a.dtype.itemsize)
    if a.size:
        _lib.lib().call("pdn_memcpy_h2d", buf.ptr, a.ctypes.data, a.nbytes, stream())
    return ndarray(buf, buf.ptr, a.shape, _contig_strides(a.shape), a.dtype)


def array(obj, dtype=None, copy=True):
    if isinstance(obj, ndarray):
        if dtype is not None and np.dtype(dtype) != obj.dtype:
            return obj.astype(dtype)
        return obj.copy() if copy else obj
    a = np.asarray(obj, dtype=dtype)
    return from_numpy(a)


def asarray(obj, dtype=None):
    return array(obj, dtype=dtype, copy=False)


def ascontiguousarray(a):
    a = asarray(a)
    return a if a.is_contiguous() else a.copy()


# ---------------------------------------------------------------------------------------
# helpers: shapes, broadcasting, views
# ---------------------------------------------------------------------------------------
def _resolve_shape(shape, size):
    shape = [int(s) for s in shape]
    if shape.count(-1) > 1:
        raise ValueError("can only specify one unknown dimension")
    if -1 in shape:
        known = -int(math.prod(shape))
        if known == 0 or size % known:
            raise ValueError(f"cannot reshape array of size {size} into shape {tuple(shape)}")
        shape[shape.index(-1)] = size // known
    if int(math.prod(shape)) != size:
        raise ValueError(f"cannot reshape array of size {size} into shape {tuple(shape)}")
    return tuple(shape)


def _reshape_strides(old_shape, old_strides, new_shape):
    """Strides of a no-copy reshape, or None when a copy is required (NumPy's rule)."""
    if int(math.prod(new_shape)) == 0:
        return _contig_strides(new_shape)
    olds = [(s, st) for s, st in zip(old_shape, old_strides) if s != 1]
    new_strides = [0] * len(new_shape)
    oi, ni = 0, 0
    on, nn = len(olds), len(new_shape)
    while ni < nn and oi < on:
        np_, op = new_shape[ni], olds[oi][0]
        nj, oj = ni + 1, oi + 1
        while np_ != op:
            if np_ < op:
                np_ *= new_shape[nj]; nj += 1
            else:
                op *= olds[oj][0]; oj += 1
        for k in range(oi, oj - 1):
            if olds[k][1] != olds[k + 1][0] * olds[k + 1][1]:
                return None
        st = olds[oj - 1][1]
        for k in range(nj - 1, ni - 1, -1):
            new_strides[k] = st
            st *= new_shape[k]
        ni, oi = nj, oj
    last = new_strides[ni - 1] if ni > 0 else 1
    for k in range(ni, nn):
        new_strides[k] = last  # trailing size-1 dims
    return tuple(new_strides)


def _broadcast(arrs, out_shape=None):
    shape = np.broadcast_shapes(*[a.shape for a in arrs]) if out_shape is None else tuple(out_shape)
    nd = len(shape)
    strides = []
    for a in arrs:
        pad = nd - a.ndim
        if pad < 0:
            raise ValueError(f"could not broadcast input array from shape {a.shape} into shape {shape}")
        st = [0] * nd
        for i, (s, k) in enumerate(zip(a.shape, a._strides)):
            if s == shape[pad + i]:
                st[pad + i] = k if s != 1 else 0
            elif s == 1:
                st[pad + i] = 0
            else:
                raise ValueError(f"could not broadcast input array from shape {a.shape} into shape {shape}")
        strides.append(st)
    return shape, strides


def _cast_into(src: ndarray, dst: ndarray):
    shape, (ss, sd) = _broadcast([src, dst], out_shape=dst.shape)
    _lib.lib().call("pdn_cast", _dtcode(src.dtype), _dtcode(dst.dtype), len(shape), _i64(shape),
                    src._ptr, _i64(ss), dst._ptr, _i64(sd), stream())


def _fill(dst: ndarray, value):
    _lib.lib().call("pdn_fill", _dtcode(dst.dtype), float(value), dst.ndim, _i64(dst.shape),
                    dst._ptr, _i64(dst._strides), stream())


def _is_scalar(o):
    return isinstance(o, (numbers.Number, np.generic, bool)) or (isinstance(o, np.ndarray) and o.ndim == 0)


def _binary(op, a, b, out=None):
    L = _lib.lib()
    code = _BOP[op]
    cmp = code >= 16
    a_s, b_s = _is_scalar(a), _is_scalar(b)
    if a_s and b_s:
        raise TypeError("hipnp binary op needs at least one device array")
    if not a_s and not isinstance(a, ndarray):
        a = asarray(a)
    if not b_s and not isinstance(b, ndarray):
        b = asarray(b)
    if a_s or b_s:
        arr, sc, mode = (b, a, 2) if a_s else (a, b, 1)
        sc_np = np.asarray(sc)
        # NumPy (NEP 50): python scalars are weak; 0-d arrays / np scalars take part in promotion
        if isinstance(sc, (int, float, bool)) and not isinstance(sc, np.generic):
            dt = arr.dtype if (arr.dtype.kind == "f" or isinstance(sc, (int, bool))) else np.result_type(arr.dtype, np.float64)
        else:
            dt = np.result_type(arr.dtype, sc_np.dtype)
        if out is not None:
            dt = out.dtype
        if arr.dtype != dt:
            arr = arr.astype(dt)
        res = out if out is not None else empty(arr.shape, np.bool_ if cmp else dt)
        shape, (sa, so) = _broadcast([arr, res], out_shape=res.shape)
        L.call("pdn_ew_binary", _dtcode(dt), code, mode, len(shape), _i64(shape), arr._ptr, _i64(sa),
               None, None, float(sc_np), res._ptr, _i64(so), stream())
        return res
    dt = np.result_type(a.dtype, b.dtype) if out is None else out.dtype
    if a.dtype != dt:
        a = a.astype(dt)
    if b.dtype != dt:
        b = b.astype(dt)
    shape = np.broadcast_shapes(a.shape, b.shape)
    res = out if out is not None else empty(shape, np.bool_ if cmp else dt)
    shape, (sa, sb, so) = _broadcast([a, b, res], out_shape=res.shape)
    L.call("pdn_ew_binary", _dtcode(dt), code, 0, len(shape), _i64(shape), a._ptr, _i64(sa),
           b._ptr, _i64(sb), 0.0, res._ptr, _i64(so), stream())
    return res


def _unary(op, a, out=None):
    a = asarray(a)
    if a.dtype.kind != "f":
        a = a.astype(np.float64)
    res = out if out is not None else empty(a.shape, a.dtype)
    _lib.lib().call("pdn_ew_unary", _dtcode(a.dtype), _UOP[op], a.ndim, _i64(a.shape), a._ptr,
                    _i64(a._strides), res._ptr, _i64(res._strides), stream())
    return res


def _norm_axes(axis, ndim):
    if axis is None:
        return tuple(range(ndim))
    if isinstance(axis, numbers.Integral):
        axis = (int(axis),)
    out = []
    for a in axis:
        a = int(a)
        if a < -ndim or a >= ndim:
            raise np.exceptions.AxisError(a, ndim)
        out.append(a % ndim)
    if len(set(out)) != len(out):
        raise ValueError("duplicate value in 'axis'")
    return tuple(out)


def _reduce(op, a, axis=None, keepdims=False):
    a = asarray(a)
    if op in ("argmax", "argmin") and axis is not None and not isinstance(axis, numbers.Integral):
        raise TypeError("argmax/argmin take a single integer axis")
    axes = _norm_axes(axis, a.ndim)
    src = a
    if a.dtype == np.bool_:
        if op != "sum":
            raise TypeError(f"HIP backend: {op} of a bool array is not supported")
        src = a.astype(np.int64)
    if src.dtype == np.int64 and op == "mean":
        src = src.astype(np.float64)
    if src.dtype == np.float16:                       # float32 accumulation, result rounded to float16
        r = _reduce(op, src.astype(np.float32), axis, keepdims)
        return r if op in ("argmax", "argmin") else r.astype(np.float16)
    flags = (ctypes.c_uint8 * _bi.max(a.ndim, 1))(*[1 if i in axes else 0 for i in range(a.ndim)])
    kept = tuple(s for i, s in enumerate(a.shape) if i not in axes)
    odt = np.int64 if op in ("argmax", "argmin") else src.dtype
    out = empty(kept, odt)
    outn = _bi.max(int(math.prod(kept)), 1)
    ws_ptr, ws_bytes = workspace(outn * 1024 * 16 + 4096 if outn <= 4096 else outn * 16 * 64)
    _lib.lib().call("pdn_reduce", _dtcode(src.dtype), _ROP[op], src.ndim, _i64(src.shape),
                    _i64(src._strides), flags, src._ptr, out._ptr, ws_ptr, ws_bytes, stream())
    if keepdims:
        out = out.reshape(tuple(1 if i in axes else s for i, s in enumerate(a.shape)))
    return out


# ---- basic / advanced indexing ------------------------------------------------------------
def _is_basic(k):
    return k is None or k is Ellipsis or isinstance(k, (slice, numbers.Integral))


def _basic_index(a: ndarray, key):
    """View for int / slice / None / Ellipsis keys; None if `key` needs a gather."""
    if not isinstance(key, tuple):
        key = (key,)
    if not all(_is_basic(k) for k in key):
        return None
    n_spec = _bi.sum(1 for k in key if k is not None and k is not Ellipsis)
    if n_spec > a.ndim:
        raise IndexError("too many indices for array")
    if _bi.sum(1 for k in key if k is Ellipsis) > 1:
        raise IndexError("an index can only have a single ellipsis")
    if Ellipsis in key:
        i = key.index(Ellipsis)
        key = key[:i] + (slice(None),) * (a.ndim - n_spec) + key[i + 1:]
    else:
        key = key + (slice(None),) * (a.ndim - n_spec)
    shape, strides, off, dim = [], [], 0, 0
    for k in key:
        if k is None:
            shape.append(1); strides.append(0)
            continue
        n, st = a.shape[dim], a._strides[dim]
        if isinstance(k, numbers.Integral):
            k = int(k)
            if k < -n or k >= n:
                raise IndexError(f"index {k} is out of bounds for axis {dim} with size {n}")
            off += (k % n) * st
        else:
            start, stop, step = k.indices(n)
            cnt = len(range(start, stop, step))
            off += start * st if cnt > 0 else 0
            shape.append(cnt); strides.append(st * step)
        dim += 1
    return a._view(shape, strides, off)


def _index_array(k):
    """int64 device array for an integer index (list / range / numpy / device)."""
    if isinstance(k, ndarray):
        return k if k.dtype == np.int64 else k.astype(np.int64)
    arr = np.asarray(k)
    if arr.dtype == np.bool_:
        raise NotImplementedError("HIP backend: boolean-mask read x[mask] has a data-dependent shape; "
                                  "use the fused sigmoid/tanh/relu ops or a multiply by the mask")
    return from_numpy(arr.astype(np.int64))


def _is_arange(k, n):
    if isinstance(k, range):
        return k == range(n)
    if isinstance(k, np.ndarray) and k.ndim == 1 and k.size == n:
        return bool(np.array_equal(k, np.arange(n)))
    return False


def _advanced_get(a: ndarray, key):
    L = _lib.lib()
    if not isinstance(key, tuple):
        key = (key,)
    # (arange(N), idx) on a 2-D float32 array: one column per row
    if (len(key) == 2 and a.ndim == 2 and a.dtype == np.float32 and _is_arange(key[0], a.shape[0])
            and not _is_basic(key[1])):
        idx = _index_array(key[1])
        if idx.shape != (a.shape[0],):
            raise IndexError("shape mismatch: indexing arrays could not be broadcast together")
        src = a if a._strides[1] == 1 else a.copy()
        out = empty((a.shape[0],), a.dtype)
        idx = ascontiguousarray(idx)
        L.call("pdn_take_cols_f32", src._ptr, a.shape[0], a.shape[1], src._strides[0],
               idx._ptr, out._ptr, err_flag_ptr(), stream())
        return out
    # integer array on axis 0 (embedding lookup), trailing basic keys applied afterwards
    if (a.ndim >= 1 and not _is_basic(key[0]) and all(_is_basic(k) for k in key[1:])
            and a.dtype == np.float32):
        idx = ascontiguousarray(_index_array(key[0]))
        inner = ndarray(a._buf, a._ptr, a.shape[1:], a._strides[1:], a.dtype)
        src = a if inner.is_contiguous() else a.copy()
        D = int(math.prod(a.shape[1:]))
        row_stride = src._strides[0] if src.shape[0] > 1 else _bi.max(D, 1)
        out = empty(idx.shape + a.shape[1:], a.dtype)
        L.call("pdn_embedding_gather_f32", src._ptr, a.shape[0], D, row_stride, idx._ptr, idx.size,
               out._ptr, err_flag_ptr(), stream())
        if len(key) > 1:
            out = out[(slice(None),) * idx.ndim + tuple(key[1:])]
        return out
    # list index on a later axis such as x[:, [-1], :]: move it to the front
    adv = [i for i, k in enumerate(key) if not _is_basic(k)]
    if len(adv) == 1 and all(isinstance(k, slice) and k == slice(None) for i, k in enumerate(key) if i != adv[0]):
        ax = adv[0]
        moved = a.transpose([ax] + [i for i in range(a.ndim) if i != ax])
        g = _advanced_get(moved.copy() if not moved.is_contiguous() else moved, (key[ax],))
        idx_nd = np.ndim(key[ax]) if not isinstance(key[ax], ndarray) else key[ax].ndim
        perm = list(range(idx_nd, idx_nd + ax)) + list(range(idx_nd)) + list(range(idx_nd + ax, g.ndim))
        return g.transpose(perm)
    raise NotImplementedError(f"HIP backend: unsupported advanced index {key!r}")


def _advanced_set(a: ndarray, key, value):
    L = _lib.lib()
    if not isinstance(key, tuple):
        key = (key,)
    value = asarray(value, dtype=a.dtype) if not isinstance(value, ndarray) else value
    if (len(key) == 2 and a.ndim == 2 and a.dtype == np.float32 and a.is_contiguous()
            and _is_arange(key[0], a.shape[0]) and not _is_basic(key[1])):
        idx = ascontiguousarray(_index_array(key[1]))
        v = ascontiguousarray(broadcast_to(value, (a.shape[0],)))
        L.call("pdn_put_cols_f32", v._ptr, idx._ptr, a._ptr, a.shape[0], a.shape[1], stream())
        return
    if len(key) == 1 and a.dtype == np.float32 and a.is_contiguous() and a.ndim >= 1:
        idx = ascontiguousarray(_index_array(key[0]))
        D = int(math.prod(a.shape[1:]))
        v = ascontiguousarray(broadcast_to(value, idx.shape + a.shape[1:]))
        ws_ptr, ws_bytes = workspace(a.shape[0] * 4)
        L.call("pdn_embedding_scatter_f32", v._ptr, idx._ptr, idx.size, a._ptr, a.shape[0], D, 0,
               None, 0.0, ws_ptr, ws_bytes, stream())
        return
    raise NotImplementedError(f"HIP backend: unsupported advanced assignment key {key!r}")


# ---------------------------------------------------------------------------------------
# module-level functions used through `xp`
# ---------------------------------------------------------------------------------------
def exp(a): return _unary("exp", a)
def log(a): return _unary("log", a)
def abs(a): return _unary("abs", a)  # noqa: A001
def sign(a): return _unary("sign", a)
def sqrt(a): return _unary("sqrt", a)
def square(a): return _unary("square", a)
def negative(a): return _unary("neg", a)
def sigmoid(a): return _unary("sigmoid", a)
def tanh(a): return _unary("tanh", a)
def maximum(a, b): return _binary("maximum", a, b)
def minimum(a, b): return _binary("minimum", a, b)
def add(a, b): return _binary("add", a, b)
def subtract(a, b): return _binary("sub", a, b)
def multiply(a, b): return _binary("mul", a, b)
def divide(a, b): return _binary("div", a, b)
def power(a, b): return _binary("pow", a, b)
def sum(a, axis=None, keepdims=False): return _reduce("sum", a, axis, keepdims)  # noqa: A001
def mean(a, axis=None, keepdims=False): return _reduce("mean", a, axis, keepdims)
def max(a, axis=None, keepdims=False): return _reduce("max", a, axis, keepdims)  # noqa: A001
def min(a, axis=None, keepdims=False): return _reduce("min", a, axis, keepdims)  # noqa: A001
def argmax(a, axis=None, keepdims=False): return _reduce("argmax", a, axis, keepdims)
def argmin(a, axis=None, keepdims=False): return _reduce("argmin", a, axis, keepdims)
def reshape(a, shape): return a.reshape(shape)
def transpose(a, axes=None): return a.transpose(axes)
def swapaxes(a, x, y): return a.swapaxes(x, y)


def expand_dims(a, axis):
    if isinstance(axis, numbers.Integral):
        axis = (axis,)
    nd = a.ndim + len(axis)
    axis = _norm_axes(axis, nd)
    it = iter(a.shape)
    return a.reshape(tuple(1 if i in axis else next(it) for i in range(nd)))


def broadcast_to(a, shape):
    a = asarray(a)
    shape = tuple(shape) if not isinstance(shape, numbers.Integral) else (int(shape),)
    _, (st,) = _broadcast([a], out_shape=shape)
    return ndarray(a._buf, a._ptr, shape, st, a.dtype)


def atleast_2d(a):
    a = asarray(a)
    if a.ndim == 0:
        return a.reshape(1, 1)
    if a.ndim == 1:
        return a.reshape(1, a.shape[0])
    return a


def concatenate(arrs, axis=0):
    arrs = [asarray(a) for a in arrs]
    nd = arrs[0].ndim
    axis = axis % nd
    dt = np.result_type(*[a.dtype for a in arrs])
    shape = list(arrs[0].shape)
    shape[axis] = _bi.sum(a.shape[axis] for a in arrs)
    out = empty(shape, dt)
    pos = 0
    for a in arrs:
        sl = [slice(None)] * nd
        sl[axis] = slice(pos, pos + a.shape[axis])
        _cast_into(a, out[tuple(sl)])
        pos += a.shape[axis]
    return out


def pad(a, pad_width, mode="constant"):
    if mode != "constant":
        raise NotImplementedError("hipnp.pad: constant mode only")
    a = asarray(a)
    shape = [s + lo + hi for s, (lo, hi) in zip(a.shape, pad_width)]
    out = zeros(shape, a.dtype)
    sl = tuple(slice(lo, lo + s) for s, (lo, hi) in zip(a.shape, pad_width))
    _cast_into(a, out[sl])
    return out


def matmul(a, b):
    """NumPy-rule matmul (1-D promotion, batched broadcast) on the fp32 MFMA GEMM."""
    a, b = asarray(a), asarray(b)
    dt = np.result_type(a.dtype, b.dtype)
    if dt == np.float16:                              # float32 MFMA, result rounded to float16
        return matmul(a.astype(np.float32), b.astype(np.float32)).astype(np.float16)
    if dt not in (np.float32, np.float64):
        raise TypeError(f"HIP backend matmul supports floating-point operands (got {a.dtype} @ {b.dtype})")
    if a.dtype != dt: a = a.astype(dt)
    if b.dtype != dt: b = b.astype(dt)
    if a.ndim == 0 or b.ndim == 0:
        raise ValueError("matmul: input operand does not have enough dimensions")
    a1, b1 = a.ndim == 1, b.ndim == 1
    A = a.reshape(1, a.shape[0]) if a1 else a
    B = b.reshape(b.shape[0], 1) if b1 else b
    M, K = A.shape[-2:]
    K2, N = B.shape[-2:]
    if K != K2:
        raise ValueError(f"matmul: Input operand 1 has a mismatch in its core dimension 0 (size {K2} is different from {K})")
    bshape = np.broadcast_shapes(A.shape[:-2], B.shape[:-2])
    out = empty(tuple(bshape) + (M, N), dt)
    gemm(A, B, out)
    if a1 and b1:
        return out.reshape(bshape)
    if a1:
        return out.reshape(tuple(bshape) + (N,))
    if b1:
        return out.reshape(tuple(bshape) + (M,))
    return out


def _collapse_batch(shape, *stride_lists):
    """Collapse broadcast batch dims into at most two (n1, n2) with per-operand strides."""
    dims = [(s, tuple(st[i] for st in stride_lists)) for i, s in enumerate(shape) if s != 1]
    merged = []
    for s, sts in dims:
        if merged and all(p == q * s for p, q in zip(merged[-1][1], sts)):
            merged[-1] = (merged[-1][0] * s, sts)
        else:
            merged.append((s, sts))
    return merged


def gemm(A, B, C, alpha=1.0, beta=0.0, bias=None, residual=None, b_colsum=None, colsum_accumulate=False):
    """C[...] = alpha * A[...] @ B[...] + bias + residual + beta * C (batch dims broadcast, views
    consumed in place).  `residual` must have C's strides; `b_colsum` (N,) receives the column
    sums of B in the same pass (only for the unbatched x^T @ g form)."""
    L = _lib.lib()
    f64 = A.dtype == np.float64 and B.dtype == np.float64 and C.dtype == np.float64
    if f64 and (bias is not None or residual is not None or b_colsum is not None):
        raise TypeError("gemm: the float64 product has no fused epilogues")
    for name, arr in (("A", A), ("B", B), ("C", C), ("bias", bias), ("residual", residual), ("b_colsum", b_colsum)):
        if arr is not None and arr.dtype != (np.float64 if f64 else np.float32):
            raise TypeError(f"gemm: operand {name} is {arr.dtype}; pdn_gemm_f32 takes float32 buffers only "
                            "(pdn_gemm_f64: all float64)")
    M, K = A.shape[-2:]
    N = B.shape[-1]
    bshape = C.shape[:-2]
    nbd = len(bshape)

    def bstr(x):
        pad = nbd - (x.ndim - 2)
        st = [0] * nbd
        for i in range(x.ndim - 2):
            st[pad + i] = x._strides[i] if x.shape[i] != 1 else 0
        return st

    if C._strides[-1] != 1 and N > 1:
        raise ValueError("gemm: output must have unit column stride")
    if residual is not None and (residual.shape != C.shape or residual._strides != C._strides):
        raise ValueError("gemm: residual must have the shape and strides of the output")
    merged = _collapse_batch(bshape, bstr(A), bstr(B), bstr(C))
    if len(merged) > 2:
        # rare: materialise operands so the batch collapses to one dim
        A2 = ascontiguousarray(broadcast_to(A, tuple(bshape) + A.shape[-2:]))
        B2 = ascontiguousarray(broadcast_to(B, tuple(bshape) + B.shape[-2:]))
        if not C.is_contiguous():
            raise ValueError("gemm: non-contiguous output with >2 batch dims")
        nb = int(math.prod(bshape))
        merged = [(nb, (M * K, K * N, M * N))]
        A, B = A2, B2
    while len(merged) < 2:
        merged.insert(0, (1, (0, 0, 0)))
    (n1, (a1, b1, c1)), (n2, (a2, b2, c2)) = merged
    if f64:
        L.call("pdn_gemm_f64", M, N, K, float(alpha), A._ptr, A._strides[-2], A._strides[-1], B._ptr,
               B._strides[-2], B._strides[-1], float(beta), C._ptr,
               C._strides[-2] if M > 1 else _bi.max(C._strides[-2], N), n1, n2, a1, a2, b1, b2, c1, c2, stream())
        return C
    # split-K scratch: offered whenever the contraction is long; capped at 256 MiB (the library
    # never splits further than the slabs it is given room for)
    ws_ptr, ws_bytes = workspace(_bi.min(L.query("pdn_gemm_f32_workspace_bytes", M, N, K, n1 * n2), 1 << 28)
                                 if (K >= 1024 and M * N * n1 * n2 <= (1 << 25)) else 0)
    ldc = C._strides[-2] if M > 1 else _bi.max(C._strides[-2], N)
    L.call("pdn_gemm_f32", M, N, K, float(alpha), A._ptr, A._strides[-2], A._strides[-1], B._ptr,
           B._strides[-2], B._strides[-1], float(beta), C._ptr, ldc,
           bias._ptr if bias is not None else None, n1, n2, a1, a2, b1, b2, c1, c2,
           residual._ptr if residual is not None else None,
           b_colsum._ptr if b_colsum is not None else None, 1 if colsum_accumulate else 0,
           ws_ptr, ws_bytes, stream())
    return C




class _Random:
    """Host-RNG initialisers (the reference seeds NumPy; init stays bit-identical, nn/init.py:29-37)."""

    @staticmethod
    def uniform(low=0.0, high=1.0, size=None):
        return from_numpy(np.asarray(np.random.uniform(low, high, size)))

    @staticmethod
    def normal(loc=0.0, scale=1.0, size=None):
        return from_numpy(np.asarray(np.random.normal(loc, scale, size)))

    @staticmethod
    def rand(*shape):
        return from_numpy(np.asarray(np.random.rand(*shape)))

    @staticmethod
    def randn(*shape):
        return from_numpy(np.asarray(np.random.randn(*shape)))


random = _Random()


# ---- parts of this module that live in files of their own (round 5) -- imported LAST: they build on the names above ----
from ._hipnp_streams import side_stream, Event, Timer, capturing, Graph                       # noqa: E402,F401
from ._hipnp_host import readback_array, _MappedHost, read_later, Mailbox, _Polled     # noqa: E402,F401
