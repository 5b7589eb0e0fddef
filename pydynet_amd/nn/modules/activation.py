"""Activation layers (surface of pydynet/nn/modules/activation.py)."""
from .module import Module
from .. import functional as F


class _Act(Module):
    def __repr__(self) -> str:
        return f"{self.__class__.__name__}()"


class Sigmoid(_Act):
    def forward(self, x): return F.sigmoid(x)


class Tanh(_Act):
    def forward(self, x): return F.tanh(x)


class ReLU(_Act):
    def forward(self, x): return F.relu(x)


class LeakyReLU(Module):
    def __init__(self, alpha: float = 0.1) -> None:
        super().__init__()
        self.alpha = float(alpha)

    def forward(self, x): return F.leaky_relu(x, self.alpha)
    def __repr__(self) -> str: return f"LeakyReLU(alpha={self.alpha})"


class Softmax(Module):
    def __init__(self, axis=None) -> None:
        super().__init__()
        self.axis = axis

    def forward(self, x): return F.softmax(x, self.axis)
    def __repr__(self) -> str: return f"Softmax(axis={self.axis})"
