"""Pooling layers (surface of pydynet/nn/modules/pool.py)."""
from .module import Module
from .. import functional as F


class _Pool(Module):
    _fn = None

    def __init__(self, kernel_size: int, stride: int, padding: int = 0) -> None:
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    def forward(self, x):
        return type(self)._fn(x, self.kernel_size, self.stride, self.padding)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding})"


class MaxPool1d(_Pool): _fn = staticmethod(F.max_pool1d)
class AvgPool1d(_Pool): _fn = staticmethod(F.avg_pool1d)
class MaxPool2d(_Pool): _fn = staticmethod(F.max_pool2d)
class AvgPool2d(_Pool): _fn = staticmethod(F.avg_pool2d)
