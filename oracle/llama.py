"""ORACLE (test infrastructure): the north-star workload restated on `oracle.tape`.

Follows /root/reference/llm/llama/model.py op for op (same node order, same parameter names
`layers.{i}.attention.Q.weight` ...): the training branch on the tape, and the eval-only KV-cache branch
(model.py:105-110) with greedy `generate` (model.py:254-269) in plain NumPy (`decode_logits`, `generate`).
Also the two smaller workloads of BASELINE.json: the MNIST-shaped MLP and the LeNet of
examples/pydynet/mnist.py:65-98 (shape-adapted to 3x32x32 as SURVEY 8(d) states).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from . import nn
from . import tape as T
from .tape import Var


def cos_sin_cache(head_dim, max_seq_len, base=10000, dtype=None):     # model.py:11-20
    inv = 1.0 / (base ** (np.arange(0, head_dim, 2)[: head_dim // 2] / head_dim))
    freqs = np.outer(np.arange(max_seq_len), inv).astype(dtype)
    return np.cos(freqs), np.sin(freqs)


def apply_rotary(xq, xk, cos, sin):                                    # model.py:23-44
    def rot(x):
        xri = x.reshape(*(x.shape[:-1] + (-1, 2)))
        r, i = xri[..., 0], xri[..., 1]
        c, s = T.unsqueeze(cos, -2), T.unsqueeze(sin, -2)
        out_r = T.unsqueeze(r * c - i * s, -1)
        out_i = T.unsqueeze(r * s + i * c, -1)
        return out_r, out_i
    qr, qi = rot(xq)
    kr, ki = rot(xk)
    q = T.concat([qr, qi], axis=-1)
    k = T.concat([kr, ki], axis=-1)
    return q.reshape(*(q.shape[:-2] + (-1,))), k.reshape(*(k.shape[:-2] + (-1,)))


class Llama:
    def __init__(self, vocab, dim, heads, ffn, max_seq, max_batch=None, layers=6, dtype=np.float32):
        self.vocab, self.dim, self.heads, self.ffn, self.n_layers = vocab, dim, heads, ffn, layers
        self.hd = dim // heads
        P = OrderedDict()
        # nn.Embedding never initialises its weight (linear.py:63-64): np.empty garbage. The
        # benchmark protocol overwrites it; zeros here only to be deterministic.
        P["tok_embedding.weight"] = nn.param(np.zeros((vocab, dim), dtype))
        cos, sin = cos_sin_cache(self.hd, max_seq, dtype=dtype)
        self.cos, self.sin = Var(cos), Var(sin)
        for i in range(layers):
            pre = f"layers.{i}."
            for n in "QKVO":                                   # model.py:80-83
                P[pre + f"attention.{n}.weight"] = nn.linear_params(dim, dim, False, dtype)["weight"]
            P[pre + "ffn.up.weight"] = nn.linear_params(dim, ffn, False, dtype)["weight"]     # :52
            P[pre + "ffn.gate.weight"] = nn.linear_params(dim, ffn, False, dtype)["weight"]   # :53
            P[pre + "ffn.down.weight"] = nn.linear_params(ffn, dim, False, dtype)["weight"]   # :54
            P[pre + "input_norm.weight"] = nn.param(np.ones(dim, dtype))
            P[pre + "post_attn_norm.weight"] = nn.param(np.ones(dim, dtype))
        P["norm.weight"] = nn.param(np.ones(dim, dtype))
        head = nn.linear_params(dim, vocab, True, dtype)       # lm_head HAS a bias (model.py:190)
        P["lm_head.weight"], P["lm_head.bias"] = head["weight"], head["bias"]
        self.params = P

    def parameters(self):
        return [p for p in self.params.values() if p.requires_grad]

    def hidden(self, ids, start_pos=0):                        # model.py:192-207
        P, H, hd = self.params, self.heads, self.hd
        B, L = ids.shape
        h = nn.embedding(ids, P["tok_embedding.weight"])
        cos, sin = self.cos[start_pos:start_pos + L], self.sin[start_pos:start_pos + L]
        mask = None
        if L > 1:
            m = np.triu(np.full((L, L), float("-inf")), k=1)
            m = np.concatenate([np.zeros((L, start_pos)), m], axis=1)
            mask = Var(m, dtype=h.dtype)
        for i in range(self.n_layers):
            pre = f"layers.{i}."
            x = nn.rmsnorm(h, P[pre + "input_norm.weight"])
            q = (x @ P[pre + "attention.Q.weight"]).reshape(B, L, H, hd)
            k = (x @ P[pre + "attention.K.weight"]).reshape(B, L, H, hd)
            v = (x @ P[pre + "attention.V.weight"]).reshape(B, L, H, hd)
            q, k = apply_rotary(q, k, cos, sin)
            q, kT = q.transpose(0, 2, 1, 3), k.transpose(0, 2, 3, 1)
            att = q @ kT / math.sqrt(hd)
            if mask is not None:
                att = att + mask
            att = nn.softmax(att, axis=-1)
            o = (att @ v.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3).reshape(B, L, -1)
            z = h + o @ P[pre + "attention.O.weight"]
            y = nn.rmsnorm(z, P[pre + "post_attn_norm.weight"])
            sw = nn.silu(y @ P[pre + "ffn.gate.weight"])
            up = y @ P[pre + "ffn.up.weight"]
            h = z + (sw * up) @ P[pre + "ffn.down.weight"]
        return nn.rmsnorm(h, P["norm.weight"])

    def logits(self, ids, start_pos=0):                        # model.py:209-211
        return nn.linear(self.hidden(ids, start_pos), self.params["lm_head.weight"], self.params["lm_head.bias"])

    def finetune_step(self, ids, targets, opt):                # model.py:226-252
        T.set_grad_enabled(True)
        opt.zero_grad()
        lg = self.logits(np.asarray(ids))
        B, L, V = lg.shape
        loss = nn.cross_entropy(lg.reshape(B * L, V), Var(np.asarray(targets).reshape(-1), dtype=np.int64))
        loss.backward()
        opt.step()
        return loss.item()


    # ---- eval branch: KV cache + greedy decoding (no tape: the reference runs it with autograd off) ----
    def reset_cache(self, max_batch, max_seq):                 # model.py:84-91: zero (max_batch, max_seq, H, hd) caches
        dt = self.params["norm.weight"].value.dtype
        self.cache = [(np.zeros((max_batch, max_seq, self.heads, self.hd), dt),
                       np.zeros((max_batch, max_seq, self.heads, self.hd), dt)) for _ in range(self.n_layers)]

    def decode_logits(self, ids, start_pos):                   # model.py:95-121, 192-211, 254-256 (eval mode)
        """Logits of the LAST position of `ids` (B, L) placed at positions [start_pos, start_pos + L)."""
        P, H, hd = {k: v.value for k, v in self.params.items()}, self.heads, self.hd
        ids = np.asarray(ids)
        B, L = ids.shape
        cos, sin = self.cos.value[start_pos:start_pos + L], self.sin.value[start_pos:start_pos + L]

        def rms(x, w):                                         # norm.py:245-248, eps 1e-6
            return x / np.sqrt((x * x).mean(-1, keepdims=True) + 1e-6) * w

        def rot(x):                                            # model.py:23-44 on interleaved pairs
            xr = x.reshape(*x.shape[:-1], -1, 2)
            r, i = xr[..., 0], xr[..., 1]
            c, s_ = cos[None, :, None, :], sin[None, :, None, :]
            return np.stack([r * c - i * s_, r * s_ + i * c], axis=-1).reshape(x.shape)

        h = P["tok_embedding.weight"][ids]
        mask = None
        if L > 1:                                              # model.py:199-203
            mask = np.concatenate([np.zeros((L, start_pos)), np.triu(np.full((L, L), float("-inf")), k=1)], axis=1)
            mask = mask.astype(h.dtype)
        for i in range(self.n_layers):
            pre = f"layers.{i}."
            x = rms(h, P[pre + "input_norm.weight"])
            q = (x @ P[pre + "attention.Q.weight"]).reshape(B, L, H, hd)
            k = (x @ P[pre + "attention.K.weight"]).reshape(B, L, H, hd)
            v = (x @ P[pre + "attention.V.weight"]).reshape(B, L, H, hd)
            q, k = rot(q), rot(k)
            ck, cv = self.cache[i]
            ck[:B, start_pos:start_pos + L] = k                # model.py:105-110
            cv[:B, start_pos:start_pos + L] = v
            k, v = ck[:B, :start_pos + L], cv[:B, :start_pos + L]
            att = q.transpose(0, 2, 1, 3) @ k.transpose(0, 2, 3, 1) / math.sqrt(hd)
            if mask is not None:
                att = att + mask
            att = np.exp(att - att.max(-1, keepdims=True))     # functional.py:43-49
            att = att / att.sum(-1, keepdims=True)
            o = (att @ v.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3).reshape(B, L, -1)
            z = h + o @ P[pre + "attention.O.weight"]
            y = rms(z, P[pre + "post_attn_norm.weight"])
            g = y @ P[pre + "ffn.gate.weight"]
            h = z + (g / (1 + np.exp(-g)) * (y @ P[pre + "ffn.up.weight"])) @ P[pre + "ffn.down.weight"]
        h = rms(h, P["norm.weight"])[:, [-1], :]               # model.py:254-256: only the last position
        return h @ P["lm_head.weight"] + P["lm_head.bias"]

    def generate(self, ids, max_new_tokens):                   # model.py:258-269
        ids = np.asarray(ids)
        B, L = ids.shape
        next_id = None
        for i, pos in enumerate(range(L, max_new_tokens)):
            logits = self.decode_logits(ids, 0) if i == 0 else self.decode_logits(next_id, pos)
            next_id = logits[:, -1, :].argmax(-1, keepdims=True)
            yield next_id, logits


class MLP:
    """examples/pydynet/mnist.py:65-79"""

    def __init__(self, dtype=np.float32):
        self.l1 = nn.linear_params(28 * 28, 1024, True, dtype)
        self.l2 = nn.linear_params(1024, 1024, True, dtype)
        self.l3 = nn.linear_params(1024, 10, True, dtype)

    def parameters(self):
        return [self.l1["weight"], self.l1["bias"], self.l2["weight"], self.l2["bias"],
                self.l3["weight"], self.l3["bias"]]

    def __call__(self, x):
        x = x.reshape(x.shape[0], -1)
        z1 = nn.relu(nn.linear(x, self.l1["weight"], self.l1["bias"]))
        z2 = nn.relu(nn.linear(z1, self.l2["weight"], self.l2["bias"]))
        return nn.linear(z2, self.l3["weight"], self.l3["bias"])


class LeNet:
    """examples/pydynet/mnist.py:82-98 with in_channels/spatial size as parameters."""

    def __init__(self, cin=3, hw=32, dtype=np.float32):
        self.c1 = nn.conv2d_params(cin, 20, 3, True, dtype)
        self.c2 = nn.conv2d_params(20, 50, 3, True, dtype)
        self.flat = (hw // 4) * (hw // 4) * 50
        self.f1 = nn.linear_params(self.flat, 500, True, dtype)
        self.f2 = nn.linear_params(500, 10, True, dtype)

    def parameters(self):
        return [self.c1["weight"], self.c1["bias"], self.c2["weight"], self.c2["bias"],
                self.f1["weight"], self.f1["bias"], self.f2["weight"], self.f2["bias"]]

    def __call__(self, x):
        x = nn.relu(nn.conv2d(x, self.c1["weight"], 1, 1) + self.c1["bias"])
        x = nn.max_pool2d(x, 2, 2)
        x = nn.relu(nn.conv2d(x, self.c2["weight"], 1, 1) + self.c2["bias"])
        x = nn.max_pool2d(x, 2, 2)
        x = x.reshape(-1, self.flat)
        x = nn.relu(nn.linear(x, self.f1["weight"], self.f1["bias"]))
        return nn.linear(x, self.f2["weight"], self.f2["bias"])


def train_step(model, x, y, opt):
    """examples/pydynet/mnist.py:159-166: loss -> zero_grad -> backward -> step."""
    loss = nn.cross_entropy(model(x), y)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.item()
