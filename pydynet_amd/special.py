"""Constructors (surface of pydynet/special.py:6-99).  Random ones draw from the HOST NumPy
generator and are then moved, so a seeded program initialises identically on cpu and hip."""
import numpy as np

from .core import Tensor


def _make(arr, dtype, device, requires_grad):
    return Tensor(arr, dtype=dtype, device=device, requires_grad=requires_grad)


def zeros(shape, dtype=None, device=None, requires_grad=False):
    return _make(np.zeros(shape), dtype, device, requires_grad)


def ones(shape, dtype=None, device=None, requires_grad=False):
    return _make(np.ones(shape), dtype, device, requires_grad)


def randn(*shape, dtype=None, device=None, requires_grad=False):
    return _make(np.random.randn(*shape), dtype, device, requires_grad)


def rand(*shape, dtype=None, device=None, requires_grad=False):
    return _make(np.random.rand(*shape), dtype, device, requires_grad)


def uniform(low, high, shape=None, dtype=None, device=None, requires_grad=False):
    return _make(np.random.uniform(low, high, size=shape), dtype, device, requires_grad)


def empty(shape, dtype=None, device=None, requires_grad=False):
    return _make(np.empty(shape, dtype=dtype), dtype, device, requires_grad)
