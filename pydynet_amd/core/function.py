"""Composite helpers over the core operators (surface of pydynet/core/function.py:4-259)."""
from __future__ import annotations

import numpy as np

from .tensor import Tensor, swapaxes


def sqrt(x: Tensor):
    return x ** 0.5


def square(x: Tensor):
    return x * x


def _sections(n, indices_or_sections):
    """Split points along an axis of length n (np.split rule: equal division required)."""
    if np.ndim(indices_or_sections) == 0:
        k = int(indices_or_sections)
        if k <= 0:
            raise ValueError("number sections must be larger than 0.")
        assert n % k == 0, "array split does not result in an equal division"
        step = n // k
        return [i * step for i in range(k + 1)]
    return [0] + [int(i) for i in indices_or_sections] + [n]


def split(x: Tensor, indices_or_sections, axis: int = 0):
    if not isinstance(x, Tensor):
        x = Tensor(x)
    axis = axis + x.ndim if axis < 0 else axis
    pts = _sections(x.shape[axis], indices_or_sections)
    if axis <= 2:
        pre = (slice(None),) * axis
        return [x[pre + (slice(a, b),)] for a, b in zip(pts[:-1], pts[1:])]
    moved = swapaxes(x, 0, axis)                     # function.py:160-165
    return [swapaxes(moved[a:b], axis, 0) for a, b in zip(pts[:-1], pts[1:])]


def vsplit(x, indices_or_sections): return split(x, indices_or_sections, 0)
def hsplit(x, indices_or_sections): return split(x, indices_or_sections, 1)
def dsplit(x, indices_or_sections): return split(x, indices_or_sections, 2)


def unsqueeze(x: Tensor, axis):
    """np.expand_dims as a reshape node."""
    probe = np.expand_dims(np.empty(x.shape, dtype=np.bool_), axis)
    return x.reshape(*probe.shape)


def squeeze(x: Tensor, axis=None):
    shape = x.shape
    if axis is not None:
        axes = (axis,) if isinstance(axis, int) else tuple(axis)
        for ax in axes:
            if ax >= len(shape) or ax < -len(shape):
                raise ValueError("Axis out of range")
            if shape[ax] != 1:
                raise ValueError(f"Cannot squeeze axis {ax} with size {shape[ax]}")
    return x.reshape(*np.squeeze(np.empty(shape, dtype=np.bool_), axis).shape)
