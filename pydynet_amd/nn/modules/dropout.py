"""Inverted dropout (surface of pydynet/nn/modules/dropout.py); the mask is drawn on the host."""
from .module import Module
from ...special import rand


class Dropout(Module):
    def __init__(self, p: float = 0.5) -> None:
        super().__init__()
        assert 0 <= p < 1
        self.p = p

    def forward(self, x):
        if self._train:
            mask = rand(*x.shape, device=x.device) < 1 - self.p
            return x * mask.astype(x.dtype) / (1 - self.p)
        return x

    def __repr__(self) -> str:
        return f"Dropout(p={self.p})"
