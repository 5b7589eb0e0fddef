"""Host-visible values of device arrays (one part of ``pydynet_amd.hipnp``, re-exported there): arrays whose host
value arrives by itself (`readback_array`), mapped pinned host blocks, `read_later` (a copy + event of its own instead of
a blocking copy on the compute stream) and the `Mailbox` a kernel stores into directly (greedy decode, llm/llama.py).
Split out of hipnp.py in round 5; nothing here computes."""
from __future__ import annotations

import builtins as _bi
import ctypes
import math

import numpy as np

from . import _lib
from .hipnp import ndarray, _state, stream, synchronize, _dev


class readback_array(ndarray):
    """A device array whose HOST value arrives by itself: its producer leaves it in host-visible memory
    (`Mailbox.slot`: a kernel stores straight into mapped host memory); `get()` / `item()` wait for THAT value only
    instead of synchronising the compute stream -- which may already be running later work (the next decode step,
    llm/llama.py).  Basic-index views (`a[0]`) keep the property.  Writing into the array drops the host value: it is
    an ordinary device array from then on."""

    __slots__ = ("_rb", "_host")

    def _settle(self):
        """Host value (NumPy) of the array, or None if it was invalidated."""
        rb = self._rb
        if rb is not None:
            self._rb = None
            rb._finish(self)
        return self._host

    def get(self):
        h = self._settle()
        return np.array(h) if h is not None else ndarray.get(self)

    def __getitem__(self, key):
        out = ndarray.__getitem__(self, key)
        if type(out) is ndarray and out._buf is self._buf:
            h = self._settle()
            if h is not None:
                v = readback_array(out._buf, out._ptr, out.shape, out._strides, out.dtype)
                v._rb, v._host = None, h[key]
                return v
        return out

    def _dirty(self):
        self._settle()
        self._host = None

    def __setitem__(self, key, value):
        self._dirty(); ndarray.__setitem__(self, key, value)

    def fill(self, value):
        self._dirty(); return ndarray.fill(self, value)

    def __iadd__(self, o): self._dirty(); return ndarray.__iadd__(self, o)
    def __isub__(self, o): self._dirty(); return ndarray.__isub__(self, o)
    def __imul__(self, o): self._dirty(); return ndarray.__imul__(self, o)
    def __itruediv__(self, o): self._dirty(); return ndarray.__itruediv__(self, o)


class _MappedHost:
    """Owner of a block of coherent pinned host memory mapped into the device (pdn_host_alloc_mapped)."""

    __slots__ = ("host", "ptr", "nbytes", "device", "__weakref__")

    def __init__(self, nbytes):
        h, d = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.lib().call("pdn_host_alloc_mapped", ctypes.byref(h), ctypes.byref(d), int(nbytes))
        self.host, self.ptr, self.nbytes, self.device = h.value, d.value, int(nbytes), _state["device"]

    def __del__(self):
        h, self.host = self.host, 0
        if h:
            try:
                _lib.lib().call("pdn_host_free", h)
            except Exception:                      # interpreter shutdown
                pass


class read_later:
    """Host value of a device array WITHOUT synchronising the compute stream: the copy into pinned host memory is queued
    behind the array's producers and an event behind the copy; `get()` / `item()` wait for that event only.  A training
    loop that reads step i's loss after queueing step i + 1 never lets the GPU run dry (`ndarray.get()` -- a blocking
    copy on the compute stream -- waits for everything queued so far, i.e. also for the step just launched: measured
    0.25 ms of idle GPU per 53 ms step in bench.py).  Pinned slots and events are recycled: allocating pinned memory
    waits for the device (0.45 ms when it was done per call)."""

    __slots__ = ("_slot", "_shape", "_dtype", "_nbytes", "_value", "_src")
    _free = {}                                    # device -> [(mapped host block, event), ...]
    _pool_cap = 16                                # slots kept per device; beyond that a returned slot is destroyed

    def __init__(self, a: "ndarray"):
        a = a if a.is_contiguous() else a.copy()
        L = _lib.lib()
        self._shape, self._dtype, self._value = a.shape, a.dtype, None
        self._nbytes = a.size * a.dtype.itemsize
        pool = read_later._free.setdefault(_state["device"], [])
        slot = next((x for x in pool if x[0].nbytes >= self._nbytes), None)
        if slot is not None:
            pool.remove(slot)
        else:
            ev = ctypes.c_void_p()
            L.call("pdn_event_create", ctypes.byref(ev), 0)
            slot = (_MappedHost(_bi.max(self._nbytes, 256)), ev.value)
        self._slot = slot
        self._src = a                                  # the source (or its contiguous copy) lives until the copy was waited for
        if self._nbytes:
            L.call("pdn_memcpy_d2h_async", slot[0].host, a._ptr, self._nbytes, stream())
        L.call("pdn_event_record", slot[1], stream())

    @staticmethod
    def _give_back(slot):
        pool = read_later._free.setdefault(slot[0].device, [])
        if len(pool) < read_later._pool_cap:
            pool.append(slot)
        else:
            try:
                _lib.lib().call("pdn_event_destroy", slot[1])
            except Exception:
                pass

    def get(self) -> np.ndarray:
        if self._value is None:
            mem, ev = self._slot
            _lib.lib().call("pdn_event_synchronize", ev)
            buf = (ctypes.c_char * mem.nbytes).from_address(mem.host)
            n = int(np.prod(self._shape, dtype=np.int64))
            self._value = np.frombuffer(buf, dtype=self._dtype, count=n).reshape(self._shape).copy()
            read_later._give_back(self._slot)
            self._slot = self._src = None
        return self._value

    def item(self):
        return self.get().item()

    def __del__(self):
        slot = getattr(self, "_slot", None)            # dropped unread: the slot goes back once its copy is through
        if slot is not None:
            try:
                _lib.lib().call("pdn_event_synchronize", slot[1])
                read_later._give_back(slot)
            except Exception:                          # interpreter shutdown
                pass


class Mailbox:
    """(n, *shape) int64 slots in host memory the GPU writes directly: a kernel stores slot i (system scope), the host
    reads it by polling -- no copy command, no event, nothing queued between two graph replays.  Slots start at -1
    (the kernels store non-negative values: token ids); `slot(i)` is slot i as a device array (its address is the
    mapped one: kernels may read it) whose `get()` / `item()` wait until the GPU has filled it."""

    def __init__(self, n, shape):
        self.shape = tuple(int(s) for s in shape)
        self.n, self.per = int(n), int(math.prod(self.shape))
        self._mem = _MappedHost(8 * self.n * self.per)
        buf = (ctypes.c_char * self._mem.nbytes).from_address(self._mem.host)
        self.host = np.frombuffer(buf, dtype=np.int64).reshape((self.n,) + self.shape)
        self.host[...] = -1
        self._ptr = self._mem.ptr

    def slot(self, i):
        strides, acc = [], 1
        for d in reversed(self.shape):
            strides.append(acc); acc *= d
        out = readback_array(self._mem, self._ptr + 8 * self.per * int(i), self.shape, tuple(reversed(strides)), np.int64)
        out._host = None
        out._rb = _Polled(self.host[int(i)])
        return out


class _Polled:
    __slots__ = ("view",)

    def __init__(self, view):
        self.view = view

    def _finish(self, arr):
        v, spins = self.view, 0
        while (v < 0).any():                       # the GPU's store has not landed yet
            spins += 1
            if spins == 200000:                    # far beyond any decode step: make sure the stream is still alive
                synchronize()
            elif spins > 400000:
                raise RuntimeError("Mailbox slot was never written by the GPU")
        arr._host = v.copy()
