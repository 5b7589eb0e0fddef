"""pydynet_amd -- MI355X-native compute backend behind PyDyNet's Tensor / nn / autograd surface.

    import pydynet_amd as pdn          # same names as `import pydynet as pdn`
    x = pdn.Tensor(a, dtype=np.float32, device="hip:0", requires_grad=True)

"cpu" tensors compute with NumPy (the reference's own CPU device); "hip"/"cuda" tensors live
in HBM and every operator runs a hand-written gfx950 kernel through the C ABI of
libpdnhip.so (include/pdn_hip.h).  There is no fallback from one to the other.
"""
from .core import (Tensor, add, sub, mul, div, pow, matmul, abs, sum, mean, min, max, argmax,
                   argmin, maximum, minimum, exp, log, sign, reshape, transpose, swapaxes, concat,
                   sigmoid, tanh, sqrt, square, vsplit, hsplit, dsplit, split, unsqueeze, squeeze)
from .special import zeros, ones, rand, randn, empty, uniform
from .cuda import Device
from .autograd import enable_grad, no_grad
from . import autograd, core, cuda, special, nn, optim  # noqa: F401

__all__ = ["Tensor", "add", "sub", "mul", "div", "pow", "matmul", "abs", "sum", "mean", "min", "max",
           "argmax", "argmin", "maximum", "minimum", "exp", "log", "sign", "reshape", "transpose",
           "swapaxes", "concat", "sigmoid", "tanh", "sqrt", "square", "vsplit", "hsplit", "dsplit",
           "split", "unsqueeze", "squeeze", "zeros", "ones", "rand", "randn", "empty", "uniform",
           "Device", "enable_grad", "no_grad"]
