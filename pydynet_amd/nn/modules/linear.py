"""Linear and Embedding layers (surface of pydynet/nn/modules/linear.py:12-79)."""
import math

from .module import Module
from ..parameter import Parameter
from .. import init, functional as F
from ...special import empty
from ...cuda import Device
from ...autograd import no_grad


class Linear(Module):
    """y = x @ W + b with W stored (in, out).  Init order (weight, then bias; host RNG) is part
    of the contract: a seeded program gets the reference's weights bit for bit."""

    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None) -> None:
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        where = Device(device)
        self.weight = Parameter(empty((in_features, out_features), device=where, dtype=dtype))
        self.bias = Parameter(empty(out_features, device=where, dtype=dtype)) if bias else None
        self.reset_paramters()

    def reset_paramters(self):                           # (the reference's spelling; `reset_parameters` is an alias)
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is None:
            return
        fan_in = init._calculate_fan(self.weight)[0]
        limit = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        init.uniform_(self.bias, -limit, limit)

    reset_parameters = reset_paramters

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)

    def __repr__(self) -> str:
        return f"Linear(in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None})"


class Embedding(Module):
    """Lookup table.  As in the reference the constructor does NOT initialise the weight
    (linear.py:63-64); call reset_parameters() or assign `weight.data`."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, device=None, dtype=None) -> None:
        super().__init__()
        self.num_embedding, self.embedding_dim, self.padding_idx = num_embeddings, embedding_dim, padding_idx
        self.weight = Parameter(empty((num_embeddings, embedding_dim), device=Device(device), dtype=dtype))

    def forward(self, x):
        return F.embedding(x, self.weight, self.padding_idx)

    def reset_parameters(self) -> None:
        init.normal_(self.weight)
        if self.padding_idx is not None:
            with no_grad():
                self.weight.data[self.padding_idx] = 0.
