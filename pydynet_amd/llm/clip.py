"""CLIP transformer blocks (llm/clip/model.py:8-113 of the reference) on this package's fused nodes.

Same class names, constructor signatures and registered parameter names as the reference
(`mha.QKV.weight`, `mha.O.bias`, `layer_norm1.scale`, `mlp.fc1.weight`, ...), so weights map by name.
What differs is the node count: the reference's block is ~45 generic tape nodes; here
  MultiHeadAttention  one biased GEMM for the packed QKV projection, ONE attention node reading q / k / v
                      as strided views of it (streaming kernels, hd = 64, with or without the additive
                      causal mask tensor), one biased output GEMM
  CLIPLayerNorm       one last-axis LayerNorm node (9 generic nodes in the reference)
  MLP                 GEMM, one `x * sigmoid(1.702 x)` node, GEMM
The weights of the full CLIP model (patch projection, encoders, tokenizer, image preprocessing) are not in
the reference tree (downloaded at run time), so the encoders themselves stay out of scope; these blocks
are what its attention / normalisation / MLP hot path is made of.
"""
import numpy as np

from .. import nn
from ..core import Tensor, fused, function as fn


def build_attention_mask(context_length: int):
    """Additive causal mask tensor (llm/clip/model.py:8-13): -inf above the diagonal."""
    mask = Tensor(np.triu(np.full((context_length, context_length), -np.inf, dtype=np.float32), 1), dtype=np.float32)
    mask.causal_pattern = True          # (the attention node then takes its causal flag instead of a general L x L mask)
    return mask


class MultiHeadAttention(nn.Module):
    def __init__(self, n_dim: int, n_heads: int):
        super().__init__()
        self.n_dim, self.n_heads, self.head_dim = n_dim, n_heads, n_dim // n_heads
        self.QKV = nn.Linear(n_dim, n_dim * 3, dtype=np.float32)
        self.O = nn.Linear(n_dim, n_dim, dtype=np.float32)

    def forward(self, x, mask):
        B, L, _ = x.shape
        xq, xk, xv = fn.split(self.QKV(x), 3, -1)                  # views into the packed projection
        shape = (B, L, self.n_heads, self.head_dim)
        causal = bool(getattr(mask, "causal_pattern", False)) and mask.shape == (L, L)
        if causal:
            mask = None                 # exactly the -inf upper triangle: the fused kernels' own causal flag
        elif mask is not None and mask.device != x.device:
            mask = Tensor(mask.numpy(), dtype=np.float32, device=x.device)
        ctx = fused.attention(xq.reshape(*shape), xk.reshape(*shape), xv.reshape(*shape), causal=causal, mask=mask)
        return self.O(ctx.reshape(B, L, -1))


class CLIPLayerNorm(nn.LayerNorm):
    """A conventional last-axis LayerNorm on the parameters of nn.LayerNorm (scale, shift); the running
    statistics the base class registers are not used (llm/clip/model.py:66-80)."""

    def __init__(self, normalized_shape, eps=0.000001, momentum=0.1, device=None, dtype=None):
        super().__init__(normalized_shape, eps, momentum, device, dtype)

    def forward(self, x):
        if len(self.normalized_shape) == 1 and x.shape[-1] == self.normalized_shape[0] and (
                not x.device.is_hip or (x.dtype == np.float32 and x.shape[-1] % 4 == 0 and x.shape[-1] <= 2048)):
            return fused.layer_norm(x, self.scale, self.shift, self.eps)
        mean = x.mean(axis=-1, keepdims=True)
        var = fn.square(x - mean).mean(axis=-1, keepdims=True)
        return (x - mean) / fn.sqrt(var + self.eps) * self.scale + self.shift


class MLP(nn.Module):
    def __init__(self, d_in: int, d_proj: int):
        super().__init__()
        self.d_in, self.d_proj = d_in, d_proj
        self.fc1 = nn.Linear(d_in, d_proj, dtype=np.float32)
        self.fc2 = nn.Linear(d_proj, d_in, dtype=np.float32)

    def forward(self, x):
        return self.fc2(fused.gated_sigmoid(self.fc1(x), 1.702))


class Transformer(nn.Module):
    def __init__(self, n_dim: int, n_head: int, mlp_dim: int):
        super().__init__()
        self.mha = MultiHeadAttention(n_dim, n_head)
        self.mlp = MLP(n_dim, mlp_dim)
        self.layer_norm1 = CLIPLayerNorm((n_dim,), eps=1e-5, dtype=np.float32)
        self.layer_norm2 = CLIPLayerNorm((n_dim,), eps=1e-5, dtype=np.float32)

    def forward(self, x, mask):
        x = x + self.mha(self.layer_norm1(x), mask)
        return x + self.mlp(self.layer_norm2(x))
