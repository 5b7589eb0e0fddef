"""Flat gradient storage: every parameter's `.grad` becomes a view into ONE contiguous fp32
buffer (16-byte aligned slots).  `zero_grad` is then a single fill, the multi-tensor Adam table
covers one address range, and the data-parallel wrapper all-reduces slices of the same buffer."""
import numpy as np


def flatten_gradients(params):
    """Re-point `p.grad` of each float32 parameter into a fresh flat buffer on their device.

    Returns (flat_array, offsets) with offsets in elements; accumulated gradient values are kept."""
    params = list(params)
    if not params:
        raise ValueError("flatten_gradients: no parameters")
    device = params[0].device
    offsets, total = [], 0
    for p in params:
        if p.dtype != np.float32 or p.grad is None:
            raise TypeError("flatten_gradients supports float32 parameters that require grad")
        if p.device != device:
            raise ValueError("flatten_gradients: parameters live on different devices")
        offsets.append(total)
        total += (p.size + 3) // 4 * 4
    with device:
        flat = device.xp.zeros((total,), dtype=np.float32)
    for p, off in zip(params, offsets):
        view = flat[off:off + p.size].reshape(p.shape)
        view[...] = p.grad
        p.grad = view
    return flat, offsets
