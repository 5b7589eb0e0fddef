"""The Llama of llm/llama/model.py written the way a user of the reference writes it: with the library's PLAIN operators
(own restatement of llm/llama/model.py:11-44, 47-58, 61-121, 124-150, 153-252 -- same operator order, same parameter names, so
the reference-generated fixtures address it by name).  RoPE is slices + broadcast products + concat (model.py:23-44),
attention is `matmul -> / sqrt(hd) -> + mask -> softmax -> matmul` over transposed views (model.py:112-121) with the
additive -inf mask rebuilt on the host every forward (model.py:199-203).

`pydynet_amd.llm.llama.Llama` is the same model on fused nodes; this file exists to pin -- and to time -- what somebody
gets who keeps the reference's own model code and only swaps the package (north_star: "examples/ and llm/llama drop in").
The only fused behaviour it can reach is what the operators themselves decide at node-construction time."""
import math

import numpy as np

import pydynet_amd as pdn
import pydynet_amd.nn as nn
import pydynet_amd.nn.functional as F
from pydynet_amd.core.tensor import Tensor


def rope_tables(head_dim, max_len, dtype, base=10000.0):
    inv = 1.0 / base ** (np.arange(0, head_dim, 2)[: head_dim // 2] / head_dim)
    ang = np.outer(np.arange(max_len), inv).astype(dtype)
    return Tensor(np.cos(ang)), Tensor(np.sin(ang))


def rotate_pairs(t, cos, sin):
    """(x[2i], x[2i + 1]) rotated by the position's angle i: split, rotate, re-interleave through concat."""
    pairs = t.reshape(*t.shape[:-1], -1, 2)
    even, odd = pairs[..., 0], pairs[..., 1]
    c, s = pdn.unsqueeze(cos, axis=-2), pdn.unsqueeze(sin, axis=-2)
    re = pdn.unsqueeze(even * c - odd * s, -1)
    im = pdn.unsqueeze(even * s + odd * c, -1)
    out = pdn.concat([re, im], axis=-1)
    return out.reshape(*out.shape[:-2], -1)


class PlainFFN(nn.Module):
    def __init__(self, dim, hidden, dtype):
        super().__init__()
        self.up = nn.Linear(dim, hidden, bias=False, dtype=dtype)
        self.gate = nn.Linear(dim, hidden, bias=False, dtype=dtype)
        self.down = nn.Linear(hidden, dim, bias=False, dtype=dtype)

    def forward(self, x):
        return self.down(F.silu(self.gate(x)) * self.up(x))


class PlainAttention(nn.Module):
    def __init__(self, dim, heads, dtype):
        super().__init__()
        self.heads, self.hd = heads, dim // heads
        self.Q = nn.Linear(dim, dim, bias=False, dtype=dtype)
        self.K = nn.Linear(dim, dim, bias=False, dtype=dtype)
        self.V = nn.Linear(dim, dim, bias=False, dtype=dtype)
        self.O = nn.Linear(dim, dim, bias=False, dtype=dtype)

    def forward(self, x, mask, cos, sin):
        B, L, _ = x.shape
        split = lambda t: t.reshape(B, L, self.heads, self.hd)
        q, k, v = split(self.Q(x)), split(self.K(x)), split(self.V(x))
        q, k = rotate_pairs(q, cos, sin), rotate_pairs(k, cos, sin)
        scores = q.transpose(0, 2, 1, 3) @ k.transpose(0, 2, 3, 1) / math.sqrt(self.hd)
        if mask is not None:
            scores = scores + mask
        ctx = F.softmax(scores, axis=-1) @ v.transpose(0, 2, 1, 3)
        return self.O(ctx.transpose(0, 2, 1, 3).reshape(B, L, -1))


class PlainBlock(nn.Module):
    def __init__(self, dim, heads, hidden, dtype):
        super().__init__()
        self.attention = PlainAttention(dim, heads, dtype)
        self.ffn = PlainFFN(dim, hidden, dtype)
        self.input_norm = nn.RMSNorm(dim, dtype=dtype)
        self.post_attn_norm = nn.RMSNorm(dim, dtype=dtype)

    def forward(self, x, mask, cos, sin):
        z = x + self.attention(self.input_norm(x), mask, cos, sin)
        return z + self.ffn(self.post_attn_norm(z))


class PlainLlama(nn.Module):
    def __init__(self, vocab, dim, heads, hidden, max_len, layers=6, dtype=np.float32):
        super().__init__()
        self.tok_embedding = nn.Embedding(vocab, dim, dtype=dtype)
        cos, sin = rope_tables(dim // heads, max_len, dtype)
        self.freqs_cos, self.freqs_sin = nn.Parameter(cos, False), nn.Parameter(sin, False)
        self.layers = nn.ModuleList([PlainBlock(dim, heads, hidden, dtype) for _ in range(layers)])
        self.norm = nn.RMSNorm(dim, dtype=dtype)
        self.lm_head = nn.Linear(dim, vocab, dtype=dtype)

    def forward_logits(self, ids):
        L = ids.shape[-1]
        h = self.tok_embedding(ids)
        cos, sin = self.freqs_cos[:L], self.freqs_sin[:L]
        mask = None
        if L > 1:                                           # rebuilt on the host every call, as the reference does
            mask = pdn.Tensor(np.triu(np.full((L, L), float("-inf")), k=1), device=h.device, dtype=h.dtype)
        for blk in self.layers:
            h = blk(h, mask, cos, sin)
        return self.lm_head(self.norm(h))

    def loss(self, ids, targets):
        logits = self.forward_logits(ids)
        B, L, V = logits.shape
        tgt = pdn.Tensor(np.asarray(targets).reshape(-1), dtype=np.int64, device=logits.device)
        return nn.CrossEntropyLoss()(logits.reshape(B * L, V), tgt)

    def finetune_step(self, ids, targets, opt):
        self.train(True)
        opt.zero_grad()
        loss = self.loss(ids, targets)
        loss.backward()
        opt.step()
        return loss.item()
