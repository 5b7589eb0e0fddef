"""Host-memory views shared by the emulator parts (tests/abi_emulator/__init__.py)."""
import ctypes
import math

import numpy as np

from pydynet_amd import _lib

_NP = {0: np.float32, 1: np.float64, 2: np.int64, 3: np.uint8, 4: np.int32, 5: np.float16}


def _ints(arr, n):
    return [int(arr[i]) for i in range(n)] if n else []


def view(ptr, shape, strides, dtype):
    """NumPy view of host memory at `ptr` with element strides (may be 0 or negative)."""
    dtype = np.dtype(dtype)
    shape, strides = [int(s) for s in shape], [int(s) for s in strides]
    if any(s == 0 for s in shape) or not ptr:
        return np.zeros(shape, dtype)
    lo = sum((s - 1) * st for s, st in zip(shape, strides) if st < 0)
    hi = sum((s - 1) * st for s, st in zip(shape, strides) if st > 0)
    n = hi - lo + 1
    buf = (ctypes.c_char * (n * dtype.itemsize)).from_address(int(ptr) + lo * dtype.itemsize)
    base = np.frombuffer(buf, dtype=dtype)
    return np.lib.stride_tricks.as_strided(base[-lo:], shape=shape, strides=[st * dtype.itemsize for st in strides])


def flat(ptr, n, dtype=np.float32):
    return view(ptr, (n,), (1,), dtype)
