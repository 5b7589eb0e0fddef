"""Normalisation layers (surface of pydynet/nn/modules/norm.py:9-248), reference semantics:
`LayerNorm` takes its statistics over the LEADING axes and keeps running statistics (it behaves
like a batch norm over tokens -- SURVEY 8a-21); RMSNorm normalises the trailing axes."""
import numpy as np

from .module import Module
from ..parameter import Parameter
from .. import init
from ...special import empty
from ... import core
from ...core import fused
from ...cuda import Device


class _RunningStatNorm(Module):
    def _alloc(self, shape, eps, momentum, device, dtype):
        kw = {"device": Device(device), "dtype": dtype}
        self.eps, self.momentum = eps, momentum
        self.running_mean = Parameter(empty(shape, **kw), requires_grad=False)
        self.running_var = Parameter(empty(shape, **kw), requires_grad=False)
        self.scale = Parameter(empty(shape, **kw))
        self.shift = Parameter(empty(shape, **kw))
        self.reset_parameters()

    def reset_parameters(self):
        init.zeros_(self.running_mean)
        init.ones_(self.running_var)
        init.zeros_(self.shift)
        init.ones_(self.scale)

    def _normalise(self, x, axis, keepdims):
        if (self._train and not keepdims and x.device.is_hip and x.dtype == np.float32
                and self.scale.dtype == np.float32 and x.ndim >= 2
                and tuple(axis if isinstance(axis, tuple) else (axis,)) == tuple(range(x.ndim - self.scale.ndim))
                and tuple(x.shape[x.ndim - self.scale.ndim:]) == tuple(self.scale.shape)):
            # statistics over the leading axes of a (rows, cols) view: one fused node (3 launches)
            return fused.col_norm(x, self.scale, self.shift, self.running_mean, self.running_var,
                                  self.eps, self.momentum, self.scale.size)
        if self._train:
            mean = x.mean(axis, keepdims) if keepdims else x.mean(axis)
            centred = x - mean
            sq = core.square(centred)
            var = sq.mean(axis, keepdims) if keepdims else sq.mean(axis)
            std_data = centred / core.sqrt(var + self.eps)
            self.running_mean *= (1 - self.momentum)
            self.running_mean += self.momentum * mean
            self.running_var *= (1 - self.momentum)
            self.running_var += self.momentum * var
            return std_data * self.scale + self.shift
        return (x - self.running_mean) * self.scale / core.sqrt(self.running_var + self.eps) + self.shift

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(momentum={self.momentum})"


class BatchNorm1d(_RunningStatNorm):
    def __init__(self, num_features, eps=1e-6, momentum=0.1, device=None, dtype=None) -> None:
        super().__init__()
        self.num_features = num_features
        self._alloc(num_features, eps, momentum, device, dtype)

    def forward(self, x):
        return self._normalise(x, 0, False)


class BatchNorm2d(_RunningStatNorm):
    def __init__(self, num_features, eps=1e-6, momentum=0.1, device=None, dtype=None) -> None:
        super().__init__()
        self.num_features = num_features
        self._alloc((1, num_features, 1, 1), eps, momentum, device, dtype)

    def forward(self, x):
        return self._normalise(x, (0, 2, 3), True)


class LayerNorm(_RunningStatNorm):
    def __init__(self, normalized_shape, eps=1e-6, momentum=0.1, device=None, dtype=None) -> None:
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape = tuple(normalized_shape)
        self._alloc(self.normalized_shape, eps, momentum, device, dtype)

    def forward(self, x):
        return self._normalise(x, tuple(range(x.ndim - len(self.normalized_shape))), False)


class RMSNorm(Module):
    def __init__(self, normalized_shape, eps=1e-6, device=None, dtype=None):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape = tuple(normalized_shape)
        self.sum_axis = tuple(-(i + 1) for i in range(len(self.normalized_shape)))
        self.eps = eps
        self.weight = Parameter(empty(self.normalized_shape, device=Device(device), dtype=dtype))
        self.reset_parameters()

    def reset_parameters(self):
        init.ones_(self.weight)

    def forward(self, x):
        cols = self.normalized_shape[-1]
        if (len(self.normalized_shape) == 1 and x.dtype == np.float32 and x.shape[-1] == cols
                and (not x.device.is_hip or (cols % 4 == 0 and cols <= 2048))):
            return fused.rms_norm(x, self.weight, self.eps)
        z = core.square(x).mean(self.sum_axis, keepdims=True)
        return x / core.sqrt(z + self.eps) * self.weight
