"""CLIP blocks (pydynet_amd/llm/clip.py) against the REAL reference's llm/clip/model.py:35-113 (vectors:
tools/gen_golden_r2.py -> tests/golden/clip_blocks.npz): biased multi-head attention with head_dim 64
(no mask, and with the additive causal mask tensor), last-axis LayerNorm, sigmoid-gated GELU MLP and one
whole Transformer block -- outputs, input gradient and every parameter gradient, on "cpu", on the
emulated C ABI and (``-m gpu``) on a real MI355X, where the attention node must take the streaming
kernels."""
import os

import numpy as np

import pydynet_amd as pdn
from pydynet_amd.core import fused
from pydynet_amd.core.tensor import Graph
from pydynet_amd.llm.clip import MultiHeadAttention, CLIPLayerNorm, MLP, Transformer, build_attention_mask
from tests.conftest import device_variants

G = os.path.join(os.path.dirname(__file__), "golden")
RT = 1e-4


def host(x):
    if isinstance(x, pdn.Tensor):
        return x.numpy()
    return x if isinstance(x, np.ndarray) else x.get()


def close(a, b, what):
    a, b = np.asarray(host(a)), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    assert err <= 1e-6 + RT * scale, (what, err, scale)


def _run(d, tag, module, call, dev):
    Graph.clear()
    for n, p in module._parameters.items():
        if p.requires_grad:
            p.data[...] = d[f"{tag}/p/{n}"]
    module.to(dev)
    x = pdn.Tensor(d["x"], dtype=np.float32, device=dev, requires_grad=True)
    y = call(module, x)
    close(y, d[f"{tag}/y"], tag + " y")
    w = d["w"][tuple(slice(0, s) for s in y.shape)]
    (y * pdn.Tensor(w, dtype=np.float32, device=dev)).sum().backward()
    close(x.grad, d[f"{tag}/dx"], tag + " dx")
    for n, p in module._parameters.items():
        if p.requires_grad:
            close(p.grad, d[f"{tag}/g/{n}"], f"{tag} grad {n}")
    return y


def check_clip_blocks_vs_reference(dev):
    d = np.load(os.path.join(G, "clip_blocks.npz"))
    B, L, D, H, M = 2, 40, 128, 2, 256
    kinds = []
    orig = fused.attention.forward_

    def spy(node, *a):
        out = orig(node, *a)
        kinds.append(node._kind)
        return out
    fused.attention.forward_ = spy
    try:
        np.random.seed(21)
        _run(d, "mha_nomask", MultiHeadAttention(D, H), lambda m, x: m(x, None), dev)
        mask = build_attention_mask(L)
        _run(d, "mha_causal", MultiHeadAttention(D, H), lambda m, x: m(x, mask), dev)
        _run(d, "layernorm", CLIPLayerNorm((D,), eps=1e-5, dtype=np.float32), lambda m, x: m(x), dev)
        _run(d, "mlp", MLP(D, M), lambda m, x: m(x), dev)
        _run(d, "block", Transformer(D, H, M), lambda m, x: m(x, mask), dev)
    finally:
        fused.attention.forward_ = orig
    if dev != "cpu":
        # hd = 64, L = 40 (not a whole number of 32-row tiles): the streaming kernels; the causal mask tensor of the
        # reference reaches them as the kernels' own causal flag (round 4) instead of an L x L additive mask
        assert kinds == ["stream"] * 3, kinds


device_variants(globals(), check_clip_blocks_vs_reference)
