"""The 1-layer Transformer classifier of the reference's CoLA example (examples/pydynet/transformer.py:
53-179), restated once and parameterised by the package namespace so that the SAME definition runs
on the real reference (tools/gen_golden.py) and on pydynet_amd (tests): self-attention with an
additive padding mask edited in place under no_grad, reference-semantics LayerNorm (statistics over
the leading axes + running averages), ReLU feed-forward, Embedding(padding_idx=0), sinusoidal
positions, logistic loss on +-1 labels."""
import numpy as np


def build(pdn, nn, F):
    class SelfAttention(nn.Module):
        def __init__(self, embed, heads):
            super().__init__()
            self.embed, self.heads, self.hd = embed, heads, embed // heads
            self.Q = nn.Linear(embed, embed, bias=False, dtype=np.float32)
            self.K = nn.Linear(embed, embed, bias=False, dtype=np.float32)
            self.V = nn.Linear(embed, embed, bias=False, dtype=np.float32)
            self.O = nn.Linear(embed, embed, bias=False, dtype=np.float32)

        def forward(self, x, mask):
            N, L = x.shape[0], x.shape[1]
            q = self.Q(x).reshape(N, L, self.heads, self.hd).transpose(0, 2, 1, 3)
            kT = self.K(x).reshape(N, L, self.heads, self.hd).transpose(0, 2, 3, 1)
            v = self.V(x).reshape(N, L, self.heads, self.hd).transpose(0, 2, 1, 3)
            att = q @ kT / self.hd ** .5
            if mask is not None:
                mask[mask.eq(1)] = np.float32('-inf')
                att = att + mask
            out = F.softmax(att, axis=-1) @ v
            return self.O(out.transpose(0, 2, 1, 3).reshape(N, L, -1))

    class Block(nn.Module):
        def __init__(self, embed, heads, expansion):
            super().__init__()
            self.attention = SelfAttention(embed, heads)
            self.norm1 = nn.LayerNorm(embed, dtype=np.float32)
            self.norm2 = nn.LayerNorm(embed, dtype=np.float32)
            self.feed_forward = nn.Sequential(nn.Linear(embed, expansion * embed, dtype=np.float32), nn.ReLU(),
                                              nn.Linear(expansion * embed, embed, dtype=np.float32))

        def forward(self, x, mask):
            h = self.norm1(self.attention(x, mask) + x)
            return self.norm2(self.feed_forward(h) + h)

    class Transformer(nn.Module):
        def __init__(self, embed, layers, heads, expansion, vocab, max_len):
            super().__init__()
            self.word_embedding = nn.Embedding(vocab, embed, padding_idx=0, dtype=np.float32)
            pos = np.arange(max_len)[:, None]
            div = np.exp(np.arange(0, embed, 2) * (-np.log(10000.0) / embed))
            pe = np.zeros((max_len, embed))
            pe[:, 0::2], pe[:, 1::2] = np.sin(pos * div), np.cos(pos * div)
            self.position_embedding = nn.Parameter(pdn.Tensor(pe.astype(np.float32)), False)
            self.layers = nn.ModuleList([Block(embed, heads, expansion) for _ in range(layers)])
            self.fc_out = nn.Linear(embed, 1, dtype=np.float32)

        def forward(self, x, mask):
            out = self.word_embedding(x) + self.position_embedding
            for layer in self.layers:
                out = layer(out, mask)
            return self.fc_out(out[:, 0, :])

    def construct_mask(x, padding_idx=0):
        with pdn.no_grad():
            return pdn.unsqueeze(x.eq(padding_idx), (1, 2)).astype(np.float32)     # (B, 1, 1, L)

    def loss_fn(net, ids, labels):
        out = net(ids, construct_mask(ids))
        return pdn.log(1 + pdn.exp(-labels * pdn.squeeze(out))).mean()

    return Transformer, loss_fn


CFG = dict(embed=64, layers=1, heads=4, expansion=3, vocab=40, max_len=12, batch=8, lr=5e-4, steps=3)


def make_inputs():
    rng = np.random.default_rng(7)
    ids = rng.integers(1, CFG["vocab"], (CFG["batch"], CFG["max_len"]))
    for i, n in enumerate(rng.integers(4, CFG["max_len"] + 1, CFG["batch"])):
        ids[i, n:] = 0                                            # right padding
    labels = rng.choice([-1.0, 1.0], CFG["batch"]).astype(np.float32)
    emb = (0.1 * rng.standard_normal((CFG["vocab"], CFG["embed"]))).astype(np.float32)
    emb[0] = 0.0
    return ids, labels, emb
