"""In-place initialisers (surface of pydynet/nn/init.py).  Samples are ALWAYS drawn from the
host NumPy generator -- also for GPU tensors -- so `np.random.seed(s)` gives bit-identical
weights on every device (the reference draws from cupy's RNG on GPU)."""
import math

import numpy as np

from ..autograd import no_grad
from ..core import Tensor


def calculate_gain(nonlinearity: str, param: float = None) -> float:
    table = {"linear": 1, "conv1d": 1, "conv2d": 1, "sigmoid": 1, "tanh": 5 / 3,
             "relu": math.sqrt(2.0),
             "leaky_relu": math.sqrt(2.0 / (1 + (param if param is not None else 0.01) ** 2))}
    return table[nonlinearity]


def _calculate_fan(tensor: Tensor):
    assert tensor.ndim >= 2
    fan_in, fan_out = tensor.shape[:2]
    if tensor.ndim > 2:
        field = math.prod(tensor.shape[2:])
        fan_in, fan_out = fan_in * field, fan_out * field
    return fan_in, fan_out


def _assign(tensor: Tensor, host_values) -> Tensor:
    with no_grad():
        tensor.data[...] = np.asarray(host_values).astype(tensor.dtype)
    return tensor


def uniform_(tensor, a=0., b=1.): return _assign(tensor, np.random.uniform(a, b, tensor.shape))
def normal_(tensor, mean=0., std=1.): return _assign(tensor, np.random.normal(mean, std, size=tensor.shape))


def constant_(tensor, val):
    with no_grad():
        tensor.data[...] = val
    return tensor


def ones_(tensor): return constant_(tensor, 1.)
def zeros_(tensor): return constant_(tensor, 0.)


def xavier_uniform_(tensor, gain=1.):
    fan_in, fan_out = _calculate_fan(tensor)
    bound = gain * math.sqrt(6. / (fan_in + fan_out))
    return uniform_(tensor, -bound, bound)


def xavier_normal_(tensor, gain=1.):
    fan_in, fan_out = _calculate_fan(tensor)
    return normal_(tensor, std=gain * math.sqrt(2 / (fan_in + fan_out)))


def _fan(tensor, mode):
    fan_in, fan_out = _calculate_fan(tensor)
    return {"fan_in": fan_in, "fan_out": fan_out}[mode]


def kaiming_uniform_(tensor, a=0., mode='fan_in', nonlinearity='relu'):
    bound = calculate_gain(nonlinearity, a) * math.sqrt(3. / _fan(tensor, mode))
    return uniform_(tensor, -bound, bound)


def kaiming_normal_(tensor, a=0., mode='fan_in', nonlinearity='relu'):
    return normal_(tensor, std=calculate_gain(nonlinearity, a) / math.sqrt(_fan(tensor, mode)))
