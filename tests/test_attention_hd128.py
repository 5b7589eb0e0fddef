"""Head dim 128 on the resident attention kernels (csrc/attention_hd128.hip): the shape of examples/pydynet/transformer.py
(dim 512, 4 heads; transformer.py:53-130) -- any length up to 1024, causal or not, with a (batch, key) padding mask
(transformer.py:92-96).  fused.attention against a float64 statement of llm/llama/model.py:112-121 /
transformer.py:120-128 (output and the three gradients), asserting that the RESIDENT kernels ran (`_kind`, launch
counters 9 / 10); on the emulated C ABI and (-m gpu) on MI355X."""
import ctypes
import math

import numpy as np
import pytest

import pydynet_amd as pdn
from pydynet_amd.core import fused
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants

HD = 128
CASES = [   # B, H, L, causal, padding (keys masked at the end of batch 1)
    (2, 2, 64, True, 0), (2, 2, 64, False, 0),
    (2, 4, 44, False, 7),           # the Transformer example's length: not a multiple of 32, padding mask
    (1, 2, 256, True, 0),           # one full query group, zig-zag causal schedule
    (1, 1, 300, False, 0),          # two forward groups, three backward groups, ragged last tile
    (2, 1, 100, True, 13),          # causal + padding
    (1, 1, 1, False, 0), (1, 2, 33, True, 0),
]


def _ref(q, k, v, go, causal, kb):
    q, k, v, go = (a.astype(np.float64) for a in (q, k, v, go))
    L = q.shape[1]
    s = np.einsum("blhd,bmhd->bhlm", q, k) / math.sqrt(HD)
    if kb is not None:
        s = s + kb.astype(np.float64)[:, None, None, :]
    if causal:
        s = s + np.triu(np.full((L, L), -np.inf), 1)
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    o = np.einsum("bhlm,bmhd->blhd", p, v)
    dv = np.einsum("bhlm,blhd->bmhd", p, go)
    dp = np.einsum("blhd,bmhd->bhlm", go, v)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / math.sqrt(HD)
    return o, np.einsum("bhlm,bmhd->blhd", ds, k), np.einsum("bhlm,blhd->bmhd", ds, q), dv


def _counters():
    from pydynet_amd import _lib
    buf = (ctypes.c_int64 * 24)()
    _lib.lib().call("pdn_kernel_counters", buf, 24, 1)
    return int(buf[9]), int(buf[10]), int(buf[11])


def check_head_dim_128_runs_on_the_resident_kernels(dev):
    for (B, H, L, causal, pad) in CASES:
        rng = np.random.default_rng(1000 * L + H)
        q, k, v, go = (rng.standard_normal((B, L, H, HD), dtype=np.float32) for _ in range(4))
        kb = None
        if pad:
            kb = np.zeros((B, L), np.float32); kb[B - 1, L - pad:] = -np.inf
        Graph.clear()
        Q, K, V = (pdn.Tensor(a, dtype=np.float32, device=dev, requires_grad=True) for a in (q, k, v))
        mask = None if kb is None else pdn.Tensor(kb.reshape(B, 1, 1, L), device=dev, dtype=np.float32)
        _counters()
        node = fused.attention(Q, K, V, causal=causal, start_pos=0, mask=mask)
        assert node._kind == "resident", (B, H, L, node._kind)
        (node * pdn.Tensor(go, dtype=np.float32, device=dev)).sum().backward()
        fwd, bwd, stream = _counters()
        assert (fwd, bwd, stream) == (1, 1, 0), (B, H, L, fwd, bwd, stream)
        ref = _ref(q, k, v, go, causal, kb)
        for got, want, what in zip((node.numpy(), Q.grad.get(), K.grad.get(), V.grad.get()), ref, ("o", "dq", "dk", "dv")):
            scale = max(float(np.abs(want).max()), 1e-30)
            err = float(np.abs(got.astype(np.float64) - want).max())
            assert err <= 2e-5 * scale + 1e-6, ((B, H, L, causal, pad), what, err, scale)


def check_packed_qkv_views_and_strides(dev):
    """q | k | v as the three column blocks of ONE packed projection buffer (row stride 3 dim): read through strides."""
    B, H, L = 2, 4, 44
    rng = np.random.default_rng(3)
    packed = rng.standard_normal((B, L, 3 * H * HD), dtype=np.float32)
    go = rng.standard_normal((B, L, H, HD), dtype=np.float32)
    Graph.clear()
    P = pdn.Tensor(packed, dtype=np.float32, device=dev, requires_grad=True)
    parts = [P[:, :, i * H * HD:(i + 1) * H * HD].reshape(B, L, H, HD) for i in range(3)]
    node = fused.attention(*parts, causal=False)
    assert node._kind == "resident"
    (node * pdn.Tensor(go, dtype=np.float32, device=dev)).sum().backward()
    q, k, v = (packed[:, :, i * H * HD:(i + 1) * H * HD].reshape(B, L, H, HD) for i in range(3))
    o, dq, dk, dv = _ref(q, k, v, go, False, None)
    want = np.concatenate([g.reshape(B, L, -1) for g in (dq, dk, dv)], -1)
    assert float(np.abs(node.numpy() - o).max()) <= 2e-5 * float(np.abs(o).max()) + 1e-6
    assert float(np.abs(P.grad.get() - want).max()) <= 2e-5 * float(np.abs(want).max()) + 1e-6


device_variants(globals(), check_head_dim_128_runs_on_the_resident_kernels)
device_variants(globals(), check_packed_qkv_views_and_strides)
