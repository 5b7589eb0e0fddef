"""Inverted dropout (surface of pydynet/nn/modules/dropout.py); the mask is drawn on the host."""
from .module import Module
from ...special import rand


class Dropout(Module):
    def __init__(self, p: float = 0.5) -> None:
        super().__init__()
        assert 0 <= p < 1
        self.p = p

    def __repr__(self) -> str:
        return f"Dropout(p={self.p})"

    def forward(self, x):
        if not self._train:
            return x
        keep = 1 - self.p
        kept = (rand(*x.shape, device=x.device) < keep).astype(x.dtype)      # one host draw per element, as the reference
        return x * kept / keep
