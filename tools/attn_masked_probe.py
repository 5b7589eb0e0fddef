"""`fused.attention` outside the benchmark class -- padding masks, lengths that are not multiples of 32, no causal mask
(llm/clip/model.py:35-63, examples/pydynet/transformer.py:92-96): which kernels the node picks and what a forward +
backward of the node costs; where both the resident (key bias, round 4) and the streaming kernels take the shape, both.
usage: python tools/attn_masked_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydynet_amd as pdn
from pydynet_amd import hipnp as hp
from pydynet_amd.core import fused
from pydynet_amd.core.tensor import Graph

CASES = [("CLIP vision  B 256, L 50, 12 heads, hd 64", 256, 50, 12, 64, False, False),
         ("CLIP text    B 256, L 77,  8 heads, hd 64, causal", 256, 77, 8, 64, True, False),
         ("padding mask B 128, L 128, 6 heads, hd 48", 128, 128, 6, 48, False, True),
         ("padding mask B 64,  L 256, 6 heads, hd 64", 64, 256, 6, 64, False, True)]
rng = np.random.default_rng(0)
for name, B, L, H, hd, causal, masked in CASES:
    q, k, v, w = (pdn.Tensor(rng.standard_normal((B, L, H, hd), dtype=np.float32), dtype=np.float32, device="hip:0",
                             requires_grad=True) for _ in range(4))
    mask = None
    if masked:
        m = np.zeros((B, 1, 1, L), np.float32)
        for b in range(B):
            m[b, 0, 0, L - (b % 17):] = -np.inf if b % 17 else 0.0
        mask = pdn.Tensor(m, dtype=np.float32, device="hip:0")
    out = {}
    for resident in (True, False):
        fused.attention.use_resident = resident

        def step():
            Graph.clear()
            for t in (q, k, v):
                t.zero_grad()
            node = fused.attention(q, k, v, causal=causal, mask=mask)
            (node * w).sum().backward()
            return node._kind
        kind = step(); step(); hp.synchronize()
        with hp.Timer() as t:
            for _ in range(10):
                step()
        out[kind] = t.ms / 10 * 1e3
    fused.attention.use_resident = True
    print(f"{name:54s} " + "   ".join(f"{k_} {us:8.1f} us" for k_, us in out.items()), flush=True)
