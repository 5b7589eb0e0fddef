"""Size-independent properties at the headline workload's full shapes (vocab 32000, seq 256, dim 288,
thousands of tokens), where a float64 oracle of the whole tensor would be too slow: linearity of the
vocabulary projection, rows of dlogits summing to zero with the right signs, softmax rows of the fused
attention summing to one and ignoring the future, Adam leaving parameters with zero gradients alone,
and the scatter-assign embedding gradient touching exactly the referenced rows.  A sampled subset of
rows is still checked against float64 arithmetic."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

V, D, H, L = 32000, 288, 6, 256


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_lm_head_linearity_and_sampled_rows(hip):
    rng = np.random.default_rng(1)
    T = 8192
    a = rng.standard_normal((T, D), dtype=np.float32)
    b = rng.standard_normal((T, D), dtype=np.float32)
    w = (rng.standard_normal((D, V), dtype=np.float32) * 0.05)
    A, Bm, W = hip.from_numpy(a), hip.from_numpy(b), hip.from_numpy(w)
    ya, yb, yab = hip.matmul(A, W), hip.matmul(Bm, W), hip.matmul(hip.from_numpy(a + b), W)
    lhs, rhs = yab.get(), ya.get() + yb.get()
    assert rel_err(lhs, rhs) < 2e-6                                   # linear in the activations
    rows = rng.integers(0, T, 16)
    assert rel_err(ya.get()[rows], a[rows].astype(np.float64) @ w.astype(np.float64)) < 1e-5
    # the weight-gradient product of the same layer is linear in the upstream gradient and symmetric in scale
    g = ya                                                            # reuse as an upstream gradient (T, V)
    dw1 = hip.matmul(A.T, g).get()
    dw2 = hip.matmul(A.T, g * 2.0).get()
    assert rel_err(dw2, 2.0 * dw1) < 1e-6
    cols = rng.integers(0, V, 8)
    assert rel_err(dw1[:, cols], a.T.astype(np.float64) @ ya.get()[:, cols].astype(np.float64)) < 2e-5


def test_cross_entropy_full_vocab_properties(hip):
    from pydynet_amd import _lib
    Lb = _lib.lib()
    rng = np.random.default_rng(2)
    rows = 8192
    x = (rng.standard_normal((rows, V), dtype=np.float32) * 3).astype(np.float32)
    t = rng.integers(0, V, rows)
    X, Tt = hip.from_numpy(x), hip.from_numpy(t)
    lr, lse, out, DX = hip.empty((rows,)), hip.empty((rows,)), hip.empty((1,)), hip.empty((rows, V))
    wsb = Lb.query("pdn_cross_entropy_colsum_workspace_bytes", rows, V)
    CS = hip.empty((V,))
    ws, wsb = hip.workspace(wsb)
    Lb.call("pdn_cross_entropy_fwd_bwd_f32", X._ptr, Tt._ptr, rows, V, 1, 1.0 / rows, lr._ptr, lse._ptr, out._ptr,
            DX._ptr, CS._ptr, ws, wsb, hip.err_flag_ptr(), hip.stream())
    hip.check_index_errors()
    d = DX.get()
    # softmax - onehot: every row sums to zero, the target entry is the only negative one
    assert np.abs(d.sum(1, dtype=np.float64)).max() < 1e-9 * 1e3
    assert (d[np.arange(rows), t] < 0).all()
    d[np.arange(rows), t] = 0
    assert (d >= 0).all()
    assert np.allclose(CS.get(), DX.get().sum(0, dtype=np.float64), rtol=1e-4, atol=1e-9)     # fused bias gradient
    # sampled rows against float64
    pick = rng.integers(0, rows, 32)
    x64 = x[pick].astype(np.float64)
    l64 = np.log(np.exp(x64 - x64.max(1, keepdims=True)).sum(1)) + x64.max(1)
    assert np.allclose(lse.get()[pick], l64, rtol=1e-6)
    assert np.allclose(lr.get()[pick], l64 - x64[np.arange(32), t[pick]], rtol=1e-5, atol=1e-6)
    assert abs(out.get()[0] - lr.get().astype(np.float64).mean()) < 1e-5 * abs(out.get()[0])


def test_fused_attention_full_length_properties(hip):
    from pydynet_amd import _lib
    Lb = _lib.lib()
    rng = np.random.default_rng(3)
    B, hd = 16, 48
    q = rng.standard_normal((B, L, H, hd), dtype=np.float32)
    k = rng.standard_normal((B, L, H, hd), dtype=np.float32)
    v = rng.standard_normal((B, L, H, hd), dtype=np.float32)
    Q, K = hip.from_numpy(q), hip.from_numpy(k)

    def run(vv, kk=None):
        o, lse = hip.empty((B, L, H, hd)), hip.empty((B, H, L))
        Vv, Kk = hip.from_numpy(vv), (K if kk is None else hip.from_numpy(kk))
        Lb.call("pdn_attention_fwd_f32", Q._ptr, Kk._ptr, Vv._ptr, o._ptr, lse._ptr, B, H, L, hd, H * hd, L * H * hd, H * hd, L * H * hd, 1,
                None, None, hip.stream())
        return o.get()

    ones = run(np.ones_like(v))
    assert np.allclose(ones, 1.0, rtol=0, atol=2e-6)                   # probabilities of every row sum to one
    base = run(v)
    k2, v2 = k.copy(), v.copy()
    k2[:, 200:], v2[:, 200:] = 7.0, -3.0                                # rewrite the future of positions < 200
    changed = run(v2, k2)
    assert np.array_equal(base[:, :200], changed[:, :200])              # causal: the past does not see it
    assert not np.allclose(base[:, 200:], changed[:, 200:])
    assert np.allclose(base[:, 0], v[:, 0], rtol=1e-6)                  # the first query attends to key 0 only


def test_adam_and_embedding_gradient_full_size(hip):
    from pydynet_amd.optim import Adam
    import pydynet_amd as pdn
    import pydynet_amd.nn as nn
    rng = np.random.default_rng(4)
    w = nn.Parameter(pdn.Tensor(rng.standard_normal((V, D), dtype=np.float32), dtype=np.float32, device="hip:0"))
    before = w.data.get()
    ids = rng.integers(0, V, (64, L))
    ids[0, :4] = 123                                                     # duplicates: the last write wins
    emb = nn.functional.embedding(pdn.Tensor(ids, dtype=np.int64, device="hip:0"), w, None)
    upstream = rng.standard_normal((64, L, D), dtype=np.float32)
    (emb * pdn.Tensor(upstream, dtype=np.float32, device="hip:0")).sum().backward()
    g = w.grad.get()
    used = np.zeros(V, bool); used[ids.reshape(-1)] = True
    assert not g[~used].any()                                            # untouched rows stay exactly zero
    flat_ids, flat_up = ids.reshape(-1), upstream.reshape(-1, D)
    last = {}
    for pos, tok in enumerate(flat_ids):
        last[tok] = pos
    for tok in (123, int(flat_ids[-1]), int(flat_ids[1000])):
        assert np.array_equal(g[tok], flat_up[last[tok]])                # scatter-ASSIGN, bit exact
    opt = Adam([w], lr=1e-3)
    opt.step()
    after = w.data.get()
    assert np.array_equal(after[~used], before[~used])                   # zero gradient: parameter unchanged
    moved = np.abs(after[used] - before[used])
    assert moved.max() <= 1e-3 * (1 + 1e-3) and moved.max() > 0.5e-3    # first Adam step: |u| <= lr (+ fp32 round-off of w)


def test_split_cross_entropy_statistics_full_size_properties(hip):
    """lm_head -> cross entropy at the benchmark's full shape (65536 tokens x 32000 entries) with the statistics taken
    from the two products (pdn_linear_rowmax_fwd_f32 + pdn_linear_ce_dx_deferred_f32).  Size-independent properties:
    the row maxima bound every logit of their row and are attained; a constant added to every entry of the bias shifts
    every log-sum-exp by exactly that constant and leaves the input gradient alone (softmax is shift invariant); the
    gradient of a row whose target holds (numerically) all the probability vanishes; sampled rows agree with float64."""
    from pydynet_amd import _lib
    Lb = _lib.lib()
    rng = np.random.default_rng(9)
    T = 65536
    x = rng.standard_normal((T, D), dtype=np.float32)
    w = (0.05 * rng.standard_normal((D, V))).astype(np.float32)
    b = (0.1 * rng.standard_normal(V)).astype(np.float32)
    t = rng.integers(0, V, T)
    x[7] = 40.0 * w[:, t[7]] / np.linalg.norm(w[:, t[7]])          # row 7: its target takes all the probability
    X, W, Tg = hip.from_numpy(x), hip.from_numpy(w), hip.from_numpy(t)
    parts = Lb.query("pdn_linear_rowmax_parts", T, V, D)
    res = []
    for shift in (0.0, 5.0):
        Bv = hip.from_numpy(b + np.float32(shift))
        logits, rowmax = hip.empty((T, V)), hip.empty((parts * T,))
        Lb.call("pdn_linear_rowmax_fwd_f32", X._ptr, W._ptr, Bv._ptr, logits._ptr, rowmax._ptr, T, V, D, D, V, V, hip.stream())
        dx, lse = hip.empty((T, D)), hip.empty((T,))
        ws, wsb = hip.workspace(Lb.query("pdn_linear_ce_dx_deferred_workspace_bytes", T, V, D))
        Lb.call("pdn_linear_ce_dx_deferred_f32", logits._ptr, rowmax._ptr, parts, Tg._ptr, 1.0 / T, W._ptr, dx._ptr, lse._ptr,
                T, V, D, ws, wsb, hip.stream())
        res.append((dx.get(), lse.get(), rowmax.get().reshape(parts, T).max(0), logits))
    (dx0, lse0, m0, lg0), (dx1, lse1, m1, _) = res
    rows = np.array([0, 7, 4097, 65535])
    z = x[rows].astype(np.float64) @ w.astype(np.float64) + b
    assert np.allclose(m0[rows], z.max(-1), rtol=1e-6, atol=1e-5)
    chunk = lg0[:2048].get()
    assert (chunk <= m0[:2048, None] + 1e-6).all() and np.allclose(chunk.max(-1), m0[:2048], rtol=0, atol=0)
    assert np.allclose(lse1 - lse0, 5.0, atol=2e-5) and np.allclose(m1 - m0, 5.0, atol=2e-5)
    assert rel_err(dx1, dx0) < 2e-6
    ref_lse = np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1)) + z.max(-1)
    assert np.allclose(lse0[rows], ref_lse, rtol=1e-6, atol=1e-5)
    p = np.exp(z - ref_lse[:, None])
    p[np.arange(len(rows)), t[rows]] -= 1.0
    assert rel_err(dx0[rows], (p / T) @ w.astype(np.float64).T) < 2e-5
    assert np.abs(dx0[7]).max() < 1e-9                      # p(target) = 1: (softmax - onehot) W^T = 0
