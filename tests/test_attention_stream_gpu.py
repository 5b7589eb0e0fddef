"""General streaming attention kernels (csrc/attention_stream.hip) against a float64 NumPy statement of
the reference's composition -- `q k^T / sqrt(hd) (+ mask) -> softmax(-1) -> @ v` and its gradients
(llm/llama/model.py:112-121, llm/clip/model.py:47-63, examples/pydynet/transformer.py:70-92) -- over
every supported head dim, ragged lengths, Lq != Lk with a start position (KV-cache prefill), additive
masks in the layouts the reference builds ((L, L) causal tensor, (B, 1, 1, L) padding mask), strided
q / k / v views, and RoPE applied in the loads.  Tolerance 2e-5 of the tensor's largest entry."""
import math

import numpy as np
import pytest


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

pytestmark = pytest.mark.gpu


def _ref(q, k, v, do, causal, start, mask, cos=None, sin=None):
    q, k, v, do = (a.astype(np.float64) for a in (q, k, v, do))          # (B, L, H, hd)
    B, Lq, H, hd = q.shape
    Lk = k.shape[1]

    def rot(a, sign):
        if cos is None:
            return a
        L = a.shape[1]
        c, s = cos[:L].astype(np.float64)[None, :, None, :], sign * sin[:L].astype(np.float64)[None, :, None, :]
        out = np.empty_like(a)
        out[..., 0::2] = a[..., 0::2] * c - a[..., 1::2] * s
        out[..., 1::2] = a[..., 0::2] * s + a[..., 1::2] * c
        return out
    qr, kr = rot(q, 1.0), rot(k, 1.0)
    Q, K, V, DO = (a.transpose(0, 2, 1, 3) for a in (qr, kr, v, do))
    s = Q @ K.swapaxes(-1, -2) / math.sqrt(hd)
    if causal:
        qi, ki = np.arange(Lq)[:, None], np.arange(Lk)[None, :]
        s = s + np.where(ki > qi + start, -np.inf, 0.0)
    if mask is not None:
        s = s + mask.astype(np.float64)
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    l = e.sum(-1, keepdims=True)
    p = e / l
    o = p @ V
    dv = p.swapaxes(-1, -2) @ DO
    dp = DO @ V.swapaxes(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / math.sqrt(hd)
    dq, dk = ds @ K, ds.swapaxes(-1, -2) @ Q
    tr = lambda a: a.transpose(0, 2, 1, 3)
    return tr(o), (m + np.log(l))[..., 0], rot(tr(dq), -1.0), rot(tr(dk), -1.0), tr(dv)


def _close(a, b, what):
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a.astype(np.float64) - b).max())
    assert err <= 2e-5 * scale + 1e-7, (what, err, scale)


CASES = [  # B, H, Lq, Lk, hd, causal, start, mask kind, rope
    (2, 2, 64, 64, 48, 1, 0, None, False),
    (1, 3, 300, 300, 48, 1, 0, None, True),         # > 256: streams, ragged tail, RoPE in the loads
    (2, 2, 40, 40, 64, 0, 0, None, False),          # CLIP image encoder: non-causal, hd 64
    (2, 2, 40, 40, 64, 0, 0, "LL", False),          # CLIP text encoder: causal mask TENSOR (L, L)
    (3, 4, 44, 44, 128, 0, 0, "pad", False),        # transformer example: hd 128, padding mask (B,1,1,L)
    (2, 2, 7, 39, 32, 1, 32, None, False),          # KV-cache prefill: 7 new queries at position 32
    (1, 1, 1, 17, 24, 1, 16, None, False),          # single query
    (1, 2, 513, 513, 96, 1, 0, None, True),         # long, hd 96
    (2, 1, 130, 130, 128, 1, 0, "pad", True),
    (1, 2, 33, 65, 24, 0, 0, None, False),
    (8, 4, 12, 12, 16, 0, 0, "pad", False),         # the CoLA example as scripted: embed 64, 4 heads
]


@pytest.mark.parametrize("case", CASES)
def test_stream_attention_matches_float64(hip, case):
    from pydynet_amd import _lib
    L = _lib.lib()
    B, H, Lq, Lk, hd, causal, start, mkind, rope = case
    rng = np.random.default_rng(hash(case[:5]) % 2 ** 31)
    mk = lambda n: rng.standard_normal((B, n, H, hd), dtype=np.float32)
    q, k, v, do = mk(Lq), mk(Lk), mk(Lk), mk(Lq)
    mask = None
    if mkind == "LL":
        mask = np.triu(np.full((Lq, Lk), -np.inf, np.float32), 1)
    elif mkind == "pad":
        mask = np.zeros((B, 1, 1, Lk), np.float32)
        for b in range(B):
            mask[b, 0, 0, Lk - 1 - 3 * b:] = -np.inf            # trailing padded keys (never all of them)
    cos = sin = None
    if rope:
        inv = 1.0 / (10000 ** (np.arange(0, hd, 2) / hd))
        fr = np.outer(np.arange(max(Lq, Lk)), inv).astype(np.float32)
        cos, sin = np.cos(fr), np.sin(fr)
    o_r, lse_r, dq_r, dk_r, dv_r = _ref(q, k, v, do, causal, start, mask, cos, sin)
    Q, K, V, DO = (hip.from_numpy(a) for a in (q, k, v, do))
    O, LSE = hip.empty((B, Lq, H, hd)), hip.empty((B, H, Lq))
    D = H * hd
    margs = [None, 0, 0, 0, 0]
    if mask is not None:
        Mdev = hip.from_numpy(mask)
        shape = (1,) * (4 - mask.ndim) + mask.shape
        st = [0 if s == 1 else t for s, t in zip(shape, Mdev.reshape(shape)._strides)]
        margs = [Mdev._ptr] + st
    C, S = (hip.from_numpy(cos), hip.from_numpy(sin)) if rope else (None, None)
    rargs = [C._ptr, S._ptr] if rope else [None, None]
    L.call("pdn_attention_stream_fwd_f32", Q._ptr, K._ptr, V._ptr, O._ptr, LSE._ptr, B, H, Lq, Lk, hd,
           D, Lq * D, D, Lk * D, causal, start, *margs, *rargs, hip.stream())
    _close(O.get(), o_r, "o")
    _close(LSE.get(), lse_r, "lse")
    DQ, DK, DV = hip.empty(q.shape), hip.empty(k.shape), hip.empty(v.shape)
    ws, wsb = hip.workspace(L.query("pdn_attention_stream_bwd_workspace_bytes", B, H, Lq))
    L.call("pdn_attention_stream_bwd_f32", Q._ptr, K._ptr, V._ptr, O._ptr, DO._ptr, LSE._ptr, DQ._ptr, DK._ptr,
           DV._ptr, B, H, Lq, Lk, hd, D, Lq * D, D, Lk * D, causal, start, *margs, *rargs, ws, wsb, hip.stream())
    _close(DQ.get(), dq_r, "dq")
    _close(DK.get(), dk_r, "dk")
    _close(DV.get(), dv_r, "dv")


def test_stream_kernels_equal_resident_kernels_on_the_benchmark_shape(hip):
    """hd 48, L 256, causal, RoPE in the loads: the two kernel families agree (and the node-level
    switch `attention.use_resident` routes the Llama step through either)."""
    from pydynet_amd import _lib
    L = _lib.lib()
    B, H, Lq, hd = 2, 3, 256, 48
    D = H * hd
    rng = np.random.default_rng(5)
    q, k, v = (hip.from_numpy(rng.standard_normal((B, Lq, H, hd), dtype=np.float32)) for _ in range(3))
    inv = 1.0 / (10000 ** (np.arange(0, hd, 2) / hd))
    fr = np.outer(np.arange(Lq), inv).astype(np.float32)
    C, S = hip.from_numpy(np.cos(fr)), hip.from_numpy(np.sin(fr))
    o1, o2, l1, l2 = hip.empty((B, Lq, H, hd)), hip.empty((B, Lq, H, hd)), hip.empty((B, H, Lq)), hip.empty((B, H, Lq))
    L.call("pdn_attention_fwd_f32", q._ptr, k._ptr, v._ptr, o1._ptr, l1._ptr, B, H, Lq, hd, D, Lq * D, D, Lq * D, 1,
           C._ptr, S._ptr, hip.stream())
    L.call("pdn_attention_stream_fwd_f32", q._ptr, k._ptr, v._ptr, o2._ptr, l2._ptr, B, H, Lq, Lq, hd, D, Lq * D,
           D, Lq * D, 1, 0, None, 0, 0, 0, 0, C._ptr, S._ptr, hip.stream())
    assert np.allclose(o1.get(), o2.get(), rtol=1e-5, atol=1e-6)
    assert np.allclose(l1.get(), l2.get(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hd,L,kind", [(32, 40, "stream"), (64, 40, "stream"), (64, 96, "resident")])
def test_attention_node_on_strided_views_and_cache_prefill(hip, hd, L, kind):
    """q / k / v that are VIEWS (packed QKV projection, llm/clip/model.py:44-46; KV cache slices,
    llm/llama/model.py:105-110): the streaming kernels, or -- head dim 64 and a whole number of 32-row tiles -- the
    resident ones reading the views through their strides; both match the GEMM + softmax composition of the same node."""
    import pydynet_amd as pdn
    from pydynet_amd.core import fused
    from pydynet_amd.core.tensor import Graph
    rng = np.random.default_rng(2)
    B, H = 2, 2
    qkv_np = rng.standard_normal((B, L, 3 * H * hd), dtype=np.float32)
    w_np = rng.standard_normal((B, L, H, hd), dtype=np.float32)
    res = []
    for flash in (True, False):
        Graph.clear()
        fused.attention.use_flash = flash
        try:
            qkv = pdn.Tensor(qkv_np, dtype=np.float32, device="hip:0", requires_grad=True)
            parts = pdn.split(qkv, 3, -1)
            q, k, v = (p.reshape(B, L, H, hd) for p in parts)
            node = fused.attention(q, k, v, causal=False)
            assert (node._kind == kind) == flash, node._kind
            (node * pdn.Tensor(w_np, dtype=np.float32, device="hip:0")).sum().backward()
            res.append((node.numpy(), qkv.grad.get()))
        finally:
            fused.attention.use_flash = True
    for a, b in zip(res[0], res[1]):
        assert np.allclose(a, b, rtol=1e-4, atol=1e-5 * np.abs(b).max())


@pytest.mark.gpu
@pytest.mark.parametrize("L,hd", [(256, 48), (512, 48), (640, 64)])
def test_resident_attention_rope_in_the_loads_packed_projection(hip, L, hd):
    """The resident kernels as fused.qkv_attention drives them: q | k | v are column blocks of ONE packed projection
    buffer (row stride 3 D), RoPE (llm/llama/model.py:23-44) is applied to q and k as they are loaded -- at the
    positions of the 256-row chunk being staged when L > 256 -- and dq, dk are rotated back as they are stored.
    Against float64 on explicitly rotated operands."""
    import math
    from pydynet_amd import _lib
    Lb = _lib.lib()
    B, H = 2, 3
    D = H * hd
    rng = np.random.default_rng(L + hd)
    qkv = rng.standard_normal((B * L, 3 * D), dtype=np.float32)
    do = rng.standard_normal((B, L, H, hd), dtype=np.float32)
    inv = 1.0 / (10000 ** (np.arange(0, hd, 2) / hd))
    fr = np.outer(np.arange(L), inv).astype(np.float32)
    cos, sin = np.cos(fr), np.sin(fr)
    QKV, DO, C, S = hip.from_numpy(qkv), hip.from_numpy(do), hip.from_numpy(cos), hip.from_numpy(sin)
    q, k, v = QKV._ptr, QKV._ptr + 4 * D, QKV._ptr + 8 * D
    o, lse, dqkv = hip.empty((B, L, H, hd)), hip.empty((B, H, L)), hip.empty((B * L, 3 * D))
    ws, wsb = hip.workspace(4 * B * H * L)
    Lb.call("pdn_attention_fwd_f32", q, k, v, o._ptr, lse._ptr, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, 1,
            C._ptr, S._ptr, hip.stream())
    Lb.call("pdn_attention_bwd_f32", q, k, v, o._ptr, DO._ptr, lse._ptr, dqkv._ptr, dqkv._ptr + 4 * D, dqkv._ptr + 8 * D,
            B, H, L, hd, 3 * D, L * 3 * D, D, L * D, 1, C._ptr, S._ptr, ws, wsb, hip.stream())

    def rot(a, sign):                                       # (B, L, H, hd) float64
        c, s = cos[None, :, None, :].astype(np.float64), sign * sin[None, :, None, :].astype(np.float64)
        out = np.empty_like(a)
        out[..., 0::2] = a[..., 0::2] * c - a[..., 1::2] * s
        out[..., 1::2] = a[..., 0::2] * s + a[..., 1::2] * c
        return out
    x = qkv.astype(np.float64).reshape(B, L, 3, H, hd)
    q64, k64, v64 = rot(x[:, :, 0], 1.0), rot(x[:, :, 1], 1.0), x[:, :, 2]
    qh, kh, vh, gh = (a.transpose(0, 2, 1, 3) for a in (q64, k64, v64, do.astype(np.float64)))
    s = qh @ kh.swapaxes(-1, -2) / math.sqrt(hd) + np.triu(np.full((L, L), -np.inf), 1)
    e = np.exp(s - s.max(-1, keepdims=True))
    p = e / e.sum(-1, keepdims=True)
    assert rel_err(o.get(), (p @ vh).transpose(0, 2, 1, 3)) < 2e-5
    dp = gh @ vh.swapaxes(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / math.sqrt(hd)
    want = np.stack([rot((ds @ kh).transpose(0, 2, 1, 3), -1.0), rot((ds.swapaxes(-1, -2) @ qh).transpose(0, 2, 1, 3), -1.0),
                     (p.swapaxes(-1, -2) @ gh).transpose(0, 2, 1, 3)], axis=2).reshape(B * L, 3 * D)
    got = dqkv.get()
    for j, name in enumerate(("dq", "dk", "dv")):
        assert rel_err(got[:, j * D:(j + 1) * D], want[:, j * D:(j + 1) * D]) < 5e-5, name
