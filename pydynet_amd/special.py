"""Constructors (surface of pydynet/special.py:6-99).  Random ones draw from the HOST NumPy
generator and are then moved, so a seeded program initialises identically on cpu and hip."""
import numpy as np

from .core import Tensor


def _make(arr, dtype, device, requires_grad):
    return Tensor(arr, dtype=dtype, device=device, requires_grad=requires_grad)


def _filled(value, shape, dtype, device, requires_grad):
    """zeros / ones on a HIP device are filled THERE (no host array, no host->device copy: such tensors are
    created inside training steps -- initial hidden states -- and must stay capturable into a hipGraph)."""
    from .cuda import Device
    dev = device if isinstance(device, Device) else Device(device)
    if dev.is_hip:
        with dev:
            arr = dev.xp.full(shape, value, dtype=np.dtype(dtype if dtype is not None else np.float64))
        return Tensor(arr, dtype=arr.dtype, copy=False, device=dev, requires_grad=requires_grad)
    return _make(np.full(shape, float(value)), dtype, device, requires_grad)


def zeros(shape, dtype=None, device=None, requires_grad=False):
    return _filled(0.0, shape, dtype, device, requires_grad)


def ones(shape, dtype=None, device=None, requires_grad=False):
    return _filled(1.0, shape, dtype, device, requires_grad)


def randn(*shape, dtype=None, device=None, requires_grad=False):
    return _make(np.random.randn(*shape), dtype, device, requires_grad)


def rand(*shape, dtype=None, device=None, requires_grad=False):
    return _make(np.random.rand(*shape), dtype, device, requires_grad)


def uniform(low, high, shape=None, dtype=None, device=None, requires_grad=False):
    return _make(np.random.uniform(low, high, size=shape), dtype, device, requires_grad)


def empty(shape, dtype=None, device=None, requires_grad=False):
    return _make(np.empty(shape, dtype=dtype), dtype, device, requires_grad)
