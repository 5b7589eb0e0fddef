"""The reference's attention chain built from PLAIN operators (llm/llama/model.py:112-121) is recognised link by link and
becomes one fused.attention node (pydynet_amd/core/fused/chain.py).  Checked on the emulated C ABI and (-m gpu) on MI355X:
  * output and gradients equal the unfused composition (class switch off) and a float64 NumPy statement;
  * the causal mask built on the host is recognised (`Tensor._causal_mask`), any other additive mask is passed on;
  * a link that somebody else reads (the scores, the probabilities) materialises as the ordinary operator;
  * a scale other than sqrt(head_dim), or operands on the cpu device, leave the plain operators alone."""
import math

import numpy as np

import pydynet_amd as pdn
import pydynet_amd.nn.functional as F
from pydynet_amd.core import fused
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants

B, L, H, HD = 2, 32, 2, 48


def _inputs(seed=0):
    rng = np.random.default_rng(seed)
    q, k, v = (rng.standard_normal((B, L, H, HD), dtype=np.float32) for _ in range(3))
    gout = rng.standard_normal((B, L, H * HD), dtype=np.float32)
    return q, k, v, gout


def _ref(q, k, v, gout, mask):
    """float64: output (B, L, H hd) and gradients of sum(out * gout)."""
    q, k, v, gout = (a.astype(np.float64) for a in (q, k, v, gout))
    s = np.einsum("blhd,bmhd->bhlm", q, k) / math.sqrt(HD)
    if mask is not None:
        s = s + mask
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    o = np.einsum("bhlm,bmhd->blhd", p, v)
    go = gout.reshape(B, L, H, HD)
    dv = np.einsum("bhlm,blhd->bmhd", p, go)
    dp = np.einsum("blhd,bmhd->bhlm", go, v)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / math.sqrt(HD)
    return o.reshape(B, L, -1), np.einsum("bhlm,bmhd->blhd", ds, k), np.einsum("bhlm,blhd->bmhd", ds, q), dv


def _chain(dev, q, k, v, gout, mask_np, scale=None, peek=None):
    Graph.clear()
    Q, K, V = (pdn.Tensor(a, dtype=np.float32, device=dev, requires_grad=True) for a in (q, k, v))
    s = Q.transpose(0, 2, 1, 3) @ K.transpose(0, 2, 3, 1) / (math.sqrt(HD) if scale is None else scale)
    if mask_np is not None:
        s = s + pdn.Tensor(mask_np, device=dev, dtype=np.float32)
    extra = None
    if peek == "scores":
        extra = s.numpy()                                   # somebody reads the scores
    p = F.softmax(s, axis=-1)
    if peek == "probs":
        extra = p.numpy()
    out = (p @ V.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3).reshape(B, L, -1)
    (out * pdn.Tensor(gout, dtype=np.float32, device=dev)).sum().backward()
    return out.numpy(), Q.grad.get() if dev != "cpu" else Q.grad, K.grad.get() if dev != "cpu" else K.grad, \
        V.grad.get() if dev != "cpu" else V.grad, extra


def _close(a, b, what, tol=2e-5):
    scale = max(float(np.abs(b).max()), 1e-30)
    assert float(np.abs(np.asarray(a, np.float64) - b).max()) <= tol * scale + 1e-7, what


def check_chain_becomes_one_attention_node(dev):
    q, k, v, gout = _inputs()
    causal = np.triu(np.full((L, L), float("-inf")), k=1)
    padding = np.zeros((B, 1, 1, L), np.float32); padding[1, ..., L - 5:] = -np.inf
    for name, mask in (("causal", causal), ("none", None), ("padding", padding)):
        n0 = fused.attn_link.fused_built
        got = _chain(dev, q, k, v, gout, mask)
        assert fused.attn_link.fused_built == n0 + 1, name                       # the chain was fused ...
        fused.attn_link.enabled = False
        try:
            plain = _chain(dev, q, k, v, gout, mask)                             # ... and equals the plain operators
        finally:
            fused.attn_link.enabled = True
        assert fused.attn_link.fused_built == n0 + 1
        ref = _ref(q, k, v, gout, mask)
        for g, p_, r, what in zip(got[:4], plain[:4], ref, ("out", "dq", "dk", "dv")):
            _close(g, r, (name, what, "vs float64"))
            _close(g, p_.astype(np.float64), (name, what, "vs plain operators"))
    t = pdn.Tensor(causal, device=dev, dtype=np.float32)
    assert t._causal_mask and not pdn.Tensor(padding, device=dev)._causal_mask
    assert not pdn.Tensor(np.triu(np.full((L, L), float("-inf")), k=2), device=dev)._causal_mask


def check_links_read_by_somebody_else_materialise(dev):
    q, k, v, gout = _inputs(1)
    causal = np.triu(np.full((L, L), float("-inf")), k=1)
    ref = _ref(q, k, v, gout, causal)
    for peek in ("scores", "probs"):
        n0 = fused.attn_link.fused_built
        got = _chain(dev, q, k, v, gout, causal, peek=peek)
        assert fused.attn_link.fused_built == n0, peek                           # not fused: the link was needed
        for g, r, what in zip(got[:4], ref, ("out", "dq", "dk", "dv")):
            _close(g, r, (peek, what))
        s = np.einsum("blhd,bmhd->bhlm", q.astype(np.float64), k.astype(np.float64)) / math.sqrt(HD) + causal
        if peek == "scores":
            fin = np.isfinite(s)
            assert np.array_equal(np.isneginf(got[4]), ~fin) and np.allclose(got[4][fin], s[fin], rtol=1e-5, atol=1e-5)
        else:
            pr = np.exp(s - s.max(-1, keepdims=True)); pr /= pr.sum(-1, keepdims=True)
            assert np.allclose(got[4], pr, rtol=1e-5, atol=1e-6)
    # another scale than sqrt(head_dim): the plain operators
    n0 = fused.attn_link.fused_built
    got = _chain(dev, q, k, v, gout, None, scale=3.0)
    assert fused.attn_link.fused_built == n0
    s = np.einsum("blhd,bmhd->bhlm", q.astype(np.float64), k.astype(np.float64)) / 3.0
    pr = np.exp(s - s.max(-1, keepdims=True)); pr /= pr.sum(-1, keepdims=True)
    _close(got[0], np.einsum("bhlm,bmhd->blhd", pr, v.astype(np.float64)).reshape(B, L, -1), "scale 3")


def test_cpu_device_keeps_plain_operators():
    q, k, v, gout = _inputs(2)
    n0 = fused.attn_link.fused_built
    got = _chain("cpu", q, k, v, gout, np.triu(np.full((L, L), float("-inf")), k=1))
    assert fused.attn_link.fused_built == n0
    for g, r, what in zip(got[:4], _ref(q, k, v, gout, np.triu(np.full((L, L), float("-inf")), k=1)), ("out", "dq", "dk", "dv")):
        _close(g, r, what)


device_variants(globals(), check_chain_becomes_one_attention_node)
device_variants(globals(), check_links_read_by_somebody_else_materialise)
