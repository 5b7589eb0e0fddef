"""Convolution layers (surface of pydynet/nn/modules/conv.py:11-114)."""
import math

from .module import Module
from ..parameter import Parameter
from .. import init, functional as F
from ...special import empty
from ...cuda import Device


class _ConvNd(Module):
    _kdims = 2

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True,
                 device=None, dtype=None) -> None:
        super().__init__()
        kw = {"device": Device(device), "dtype": dtype}
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.padding, self.stride = kernel_size, padding, stride
        self.weight = Parameter(empty((out_channels, in_channels) + (kernel_size,) * self._kdims, **kw))
        self.bias = Parameter(empty((1, out_channels) + (1,) * self._kdims, **kw)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan(self.weight)
            if fan_in != 0:
                init.uniform_(self.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(in_channels={self.in_channels}, out_channels={self.out_channels}, "
                f"kernel_size={self.kernel_size}, padding={self.padding}, stride={self.stride}, "
                f"bias={self.bias is not None})")


class Conv2d(_ConvNd):
    _kdims = 2

    def forward(self, x):
        # bias (1, O, 1, 1) rides in the GEMM epilogue of the fused conv node
        return F.conv2d(x, self.weight, self.padding, self.stride, bias=self.bias)


class Conv1d(_ConvNd):
    _kdims = 1

    def forward(self, x):
        out = F.conv1d(x, self.weight, self.padding, self.stride)
        return out + self.bias if self.bias is not None else out
