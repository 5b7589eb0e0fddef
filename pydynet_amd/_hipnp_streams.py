"""Streams, timing and hipGraph capture / replay (one part of ``pydynet_amd.hipnp``, re-exported there): a side stream
ordered against the compute stream by events, HIP-event timers on the launch stream, and `Graph` -- a whole training or
decode step captured inside a private allocation pool and replayed as one launch.  Split out of hipnp.py in round 5
(VERDICT round 4, item 9)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .hipnp import _state, stream, set_stream, synchronize, _dev

_side = {}        # device -> (side stream, fork event, join event)


class side_stream:
    """Run the launches of the `with` body on a second stream of the current device, concurrently with
    whatever the compute stream is given next; `join()` makes the compute stream wait for them.

        with hipnp.side_stream() as s:      # side stream waits for everything enqueued so far
            hipnp.gemm(x.T, g, dw)          # ... runs beside ...
        hipnp.gemm(g, w.T, dx)              # ... this one
        s.join()                            # later compute-stream work sees both results

    Lifetime rule (the allocator orders reuse on the compute stream only): join before any buffer the
    body touched can be released, i.e. before the enclosing operator returns.  Workspaces are per stream."""

    def __enter__(self):
        L = _lib.lib()
        dev = _state["device"]
        ent = _side.get(dev)
        if ent is None:
            st, e0, e1 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            L.call("pdn_stream_create", ctypes.byref(st), 0)
            L.call("pdn_event_create", ctypes.byref(e0), 0)
            L.call("pdn_event_create", ctypes.byref(e1), 0)
            ent = _side[dev] = (st.value, e0, e1)
        self._main = stream()
        self._side, self._e0, self._e1 = ent
        L.call("pdn_event_record", self._e0, self._main)
        L.call("pdn_stream_wait_event", self._side, self._e0)
        _state["stream"] = self._side
        return self

    def __exit__(self, *exc):
        _state["stream"] = self._main
        _lib.lib().call("pdn_event_record", self._e1, self._side)
        return False

    def join(self):
        _lib.lib().call("pdn_stream_wait_event", self._main, self._e1)


class Event:
    """One timing event of the library (pdn_event_*): `record()` on the launch stream, `a.elapsed_ms(b)` once both have
    completed (synchronises on `b`).  Destroyed with the object."""

    def __init__(self):
        self._e = ctypes.c_void_p()
        _lib.lib().call("pdn_event_create", ctypes.byref(self._e), 1)

    def record(self, on=None):
        _lib.lib().call("pdn_event_record", self._e, stream() if on is None else on)
        return self

    def elapsed_ms(self, later: "Event") -> float:
        L = _lib.lib()
        L.call("pdn_event_synchronize", later._e)
        ms = ctypes.c_float()
        L.call("pdn_event_elapsed_ms", self._e, later._e, ctypes.byref(ms))
        return float(ms.value)

    def __del__(self):
        try:
            if self._e:
                _lib.lib().call("pdn_event_destroy", self._e)
                self._e = None
        except Exception:
            pass


class Timer:
    """HIP-event stopwatch on the compute stream: `with Timer() as t: ...; t.ms`."""

    def __enter__(self):
        L = _lib.lib()
        self._e = [ctypes.c_void_p(), ctypes.c_void_p()]
        for e in self._e:
            L.call("pdn_event_create", ctypes.byref(e), 1)
        L.call("pdn_event_record", self._e[0], stream())
        return self

    def __exit__(self, *exc):
        L = _lib.lib()
        L.call("pdn_event_record", self._e[1], stream())
        L.call("pdn_event_synchronize", self._e[1])
        ms = ctypes.c_float()
        L.call("pdn_event_elapsed_ms", self._e[0], self._e[1], ctypes.byref(ms))
        self.ms = ms.value
        for e in self._e:
            L.call("pdn_event_destroy", e)
        return False


_capture = {"graph": None}


def capturing():
    """The Graph being captured right now (or warmed up inside its private pool), else None."""
    return _capture["graph"]


class Graph:
    """A whole step captured once and replayed as ONE hipGraph launch: the reference pays a Python object
    and at least one kernel launch per scalar-level operator, which bounds small-batch steps
    by launch latency; a replay costs one launch and no Python work.

        g = hipnp.Graph()
        loss = g.capture(step)          # runs `step` twice: once to fill the private pool, once captured
        for _ in range(n):
            ids.data[...] = next_batch  # refresh the static input buffers in place (optional)
            g.replay()                  # `loss` (and anything else `step` returned) is overwritten in place

    Rules for `step`: everything on the device (no `.item()`, no host arrays turned into device tensors,
    no dropout drawing host random numbers); tensors it allocates live in a pool private to the graph,
    so the arrays it returns stay valid -- and are rewritten -- across replays.  Optimizers that keep a
    host-side step counter (Adam) switch to a device-side counter while capturing."""

    def __init__(self):
        self._exec, self._pool, self.nodes, self._hooks, self._keep = None, None, 0, [], None
        self._ws, self._pinned = {}, []          # scratch buffers / side tables the captured launches point into

    def pin(self, obj):
        """Keep `obj` (an array whose raw pointer a captured launch holds: a scratch workspace, an optimizer's
        chunk table) alive until destroy(): a replay writes through the pointers baked in at capture time, so
        nothing they address may go back to the allocator while the graph can still be launched."""
        self._pinned.append(obj)
        return obj

    def on_replay(self, fn):
        """Host bookkeeping to run at every replay (e.g. an optimizer's step counter)."""
        self._hooks.append(fn)

    def capture(self, step):
        import gc
        L = _lib.lib()
        if _capture["graph"] is not None:
            raise RuntimeError("a Graph is already being captured")
        st = stream()
        pool = ctypes.c_int()
        L.call("pdn_pool_create", ctypes.byref(pool))
        self._pool = pool.value
        _capture["graph"] = self
        self._warm = True
        try:
            L.call("pdn_pool_activate", self._pool)
            out = step()                                   # fills the pool (driver allocations happen here)
            L.call("pdn_stream_synchronize", st)
            del out
            gc.collect()
            self._warm = False
            self._hooks = []
            L.call("pdn_graph_begin_capture", st)
            try:
                out = step()
            finally:
                h, n = ctypes.c_void_p(), ctypes.c_int()
                L.call("pdn_graph_end_capture", st, ctypes.byref(h), ctypes.byref(n))
            self._exec, self.nodes = h.value, n.value
        finally:
            L.call("pdn_pool_activate", 0)
            _capture["graph"] = None
        self._keep = out
        self.replay()                                      # the captured run itself executed nothing
        return out

    @property
    def warming(self):
        return getattr(self, "_warm", False)

    def replay(self):
        for fn in self._hooks:
            fn()
        _lib.lib().call("pdn_graph_launch", self._exec, stream())

    def pool_stats(self):
        vals = [ctypes.c_int64() for _ in range(3)]
        _lib.lib().call("pdn_pool_stats", self._pool, *[ctypes.byref(v) for v in vals])
        return dict(zip(("in_use", "reserved", "device_allocs"), (v.value for v in vals)))

    def destroy(self):
        L = _lib.lib()
        if self._exec:
            L.call("pdn_stream_synchronize", stream())
            L.call("pdn_graph_destroy", self._exec)
            self._exec = None
        self._keep = None
        self._ws, self._pinned = {}, []
        if self._pool:
            L.call("pdn_pool_destroy", self._pool)
            self._pool = None
