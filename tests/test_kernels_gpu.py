"""GPU parity of each C-ABI kernel against a NumPy statement of the same reference op.

Tolerances: bit-exact for index/copy work (gather, scatter-assign, strided copy, compare);
fp32 math within rtol 2e-5 / atol 2e-6 of NumPy's fp32 result for streams, and within
1e-4 relative (north_star tolerance) for GEMM-shaped reductions whose summation order
differs from OpenBLAS.
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RT, AT = 2e-5, 2e-6


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(256, 288, 288), (256, 768, 288), (256, 288, 768),
                                   (128, 32000, 288), (100, 70, 50), (1, 5, 7), (33, 1, 9),
                                   (64, 64, 27), (512, 96, 64)])
def test_gemm_nn(hip, M, N, K):
    rng = np.random.default_rng(M * 7 + N)
    a = rng.standard_normal((M, K), dtype=np.float32)
    b = rng.standard_normal((K, N), dtype=np.float32)
    c = hip.matmul(hip.from_numpy(a), hip.from_numpy(b)).get()
    ref = a.astype(np.float64) @ b.astype(np.float64)
    assert c.shape == (M, N) and c.dtype == np.float32
    assert rel_err(c, ref) < 1e-5


def test_gemm_mid_size_dispatch_edges(hip):
    """The three products of a Linear layer (x W, g W^T, x^T g) around the tile-rule boundaries of csrc/gemm.hip (round 6):
    outputs of 2^18 / 2^20 / 2^24 elements, fewer or more than 256 tiles of 128 x 128, contractions 64 / 512 / 513 / 1536 /
    4096, weight gradients whose tiled tiles pad the output (streaming kernel) or do not (tiled + k-split), extents that
    are not multiples of the tile or of four (scalar staging), accumulation into C -- each against float64."""
    rng = np.random.default_rng(77)
    shapes = [(5632, 512, 512), (5632, 512, 1536), (5632, 1536, 512), (2816, 512, 512), (4096, 3200, 500), (4096, 500, 500),
              (8192, 784, 1024), (2048, 512, 512), (2047, 513, 511), (1024, 1024, 1024), (8200, 132, 260), (3000, 64, 520),
              (6000, 4096, 192), (16384, 288, 768), (16390, 290, 770), (512, 4100, 2052)]
    from pydynet_amd import _lib
    if type(_lib.lib()).__name__ == "EmulatedLib":                          # (no dispatch to exercise: the host plumbing only)
        shapes = [(2047, 513, 511), (3000, 64, 520)]
    for R, I, O in shapes:
        x = rng.standard_normal((R, I), dtype=np.float32)
        w = rng.standard_normal((I, O), dtype=np.float32)
        g = rng.standard_normal((R, O), dtype=np.float32)
        X, W, G = hip.from_numpy(x), hip.from_numpy(w), hip.from_numpy(g)
        x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), g.astype(np.float64)
        assert rel_err(hip.matmul(X, W).get(), x64 @ w64) < 2e-5, (R, I, O, "x W")
        assert rel_err(hip.matmul(G, W.T).get(), g64 @ w64.T) < 2e-5, (R, I, O, "g W^T")
        assert rel_err(hip.matmul(X.T, G).get(), x64.T @ g64) < 2e-5, (R, I, O, "x^T g")
        c0 = rng.standard_normal((I, O), dtype=np.float32)
        C = hip.from_numpy(c0.copy())
        hip.gemm(X.T, G, C, beta=1.0)                                       # gradient accumulation through the slab pass
        assert rel_err(C.get(), x64.T @ g64 + c0) < 2e-5, (R, I, O, "x^T g + C")


def test_gemm_transposed_views_and_identity_check(hip):
    # asymmetric B with A = I catches swapped row/col in the MFMA C layout
    n = 96
    a = np.eye(n, dtype=np.float32)
    b = (np.arange(n * 80, dtype=np.float32).reshape(n, 80) % 17) - 3.0
    c = hip.matmul(hip.from_numpy(a), hip.from_numpy(b)).get()
    assert np.array_equal(c, b)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((300, 288), dtype=np.float32)
    w = rng.standard_normal((288, 768), dtype=np.float32)
    g = rng.standard_normal((300, 768), dtype=np.float32)
    X, W, G = map(hip.from_numpy, (x, w, g))
    dx = hip.matmul(G, W.T).get()          # NT
    dw = hip.matmul(X.T, G).get()          # TN
    assert rel_err(dx, g.astype(np.float64) @ w.T.astype(np.float64)) < 1e-5
    assert rel_err(dw, x.T.astype(np.float64) @ g.astype(np.float64)) < 1e-5


def test_gemm_split_k_weight_gradient(hip):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((8192, 288), dtype=np.float32)
    g = rng.standard_normal((8192, 288), dtype=np.float32)
    dw = hip.matmul(hip.from_numpy(x).T, hip.from_numpy(g)).get()
    ref = x.T.astype(np.float64) @ g.astype(np.float64)
    assert rel_err(dw, ref) < 2e-5


@pytest.mark.parametrize("T,M,N", [(8192, 288, 768), (4099, 288, 288), (2050, 200, 100), (6000, 768, 288)])
def test_gemm_weight_gradient_streaming_kernel(hip, T, M, N):
    # long-K x^T @ g: wave-streaming kernel, k-split over waves and blocks, ragged K and edges,
    # accumulation into an existing gradient (beta = 1), bit-reproducible across runs
    rng = np.random.default_rng(T + M)
    x = rng.standard_normal((T, M), dtype=np.float32)
    g = rng.standard_normal((T, N), dtype=np.float32)
    c0 = rng.standard_normal((M, N), dtype=np.float32)
    X, G = hip.from_numpy(x), hip.from_numpy(g)
    ref = x.T.astype(np.float64) @ g.astype(np.float64)
    C = hip.from_numpy(c0.copy())
    hip.gemm(X.T, G, C, beta=1.0)
    first = C.get()
    assert rel_err(first, ref + c0) < 2e-5
    C2 = hip.from_numpy(c0.copy())
    hip.gemm(X.T, G, C2, beta=1.0)
    assert np.array_equal(first, C2.get())
    C3 = hip.empty((M, N))
    hip.gemm(X.T, G, C3, alpha=0.5)
    assert rel_err(C3.get(), 0.5 * ref) < 2e-5


def test_gemm_batched_head_views(hip):
    # attention layout: (B, L, H, hd) viewed as (B, H, L, hd) without a copy
    B, L, H, hd = 2, 64, 6, 48
    rng = np.random.default_rng(2)
    q = rng.standard_normal((B, L, H, hd), dtype=np.float32)
    k = rng.standard_normal((B, L, H, hd), dtype=np.float32)
    Q, Kt = hip.from_numpy(q).transpose(0, 2, 1, 3), hip.from_numpy(k).transpose(0, 2, 3, 1)
    s = hip.matmul(Q, Kt).get()
    ref = np.matmul(q.transpose(0, 2, 1, 3).astype(np.float64), k.transpose(0, 2, 3, 1).astype(np.float64))
    assert s.shape == (B, H, L, L)
    assert rel_err(s, ref) < 1e-5
    p = rng.standard_normal((B, H, L, L), dtype=np.float32)
    o = hip.matmul(hip.from_numpy(p), hip.from_numpy(k).transpose(0, 2, 1, 3)).get()
    assert rel_err(o, np.matmul(p.astype(np.float64), k.transpose(0, 2, 1, 3).astype(np.float64))) < 1e-5


def test_gemm_bias_beta_and_1d_rules(hip):
    rng = np.random.default_rng(3)
    a = rng.standard_normal((40, 36), dtype=np.float32)
    b = rng.standard_normal((36, 52), dtype=np.float32)
    bias = rng.standard_normal(52, dtype=np.float32)
    c0 = rng.standard_normal((40, 52), dtype=np.float32)
    C = hip.from_numpy(c0.copy())
    hip.gemm(hip.from_numpy(a), hip.from_numpy(b), C, alpha=0.5, beta=2.0, bias=hip.from_numpy(bias))
    assert np.allclose(C.get(), 0.5 * (a @ b) + bias + 2.0 * c0, rtol=1e-5, atol=1e-5)
    v = rng.standard_normal(36, dtype=np.float32)
    assert np.allclose(hip.matmul(hip.from_numpy(v), hip.from_numpy(b)).get(), v @ b, rtol=1e-5, atol=1e-5)
    assert np.allclose(hip.matmul(hip.from_numpy(a), hip.from_numpy(v)).get(), a @ v, rtol=1e-5, atol=1e-5)
    d = hip.matmul(hip.from_numpy(v), hip.from_numpy(v)).get()
    assert d.shape == () and np.allclose(d, v @ v, rtol=1e-5)


# ---------------------------------------------------------------------------------------
def test_elementwise_broadcast_and_scalars(hip):
    rng = np.random.default_rng(4)
    a = rng.standard_normal((3, 1, 5, 4), dtype=np.float32)
    b = rng.standard_normal((2, 1, 4), dtype=np.float32)
    A, B = hip.from_numpy(a), hip.from_numpy(b)
    for f in ("__add__", "__sub__", "__mul__", "__truediv__"):
        got = getattr(A, f)(B).get()
        ref = getattr(a, f)(b)
        assert got.shape == ref.shape and got.dtype == ref.dtype
        assert np.allclose(got, ref, rtol=RT, atol=AT)
    assert np.array_equal((A > B).get(), a > b)
    assert np.array_equal((A == A).get(), a == a)
    assert np.allclose((A * 2.5).get(), a * 2.5, rtol=RT)
    assert np.allclose((1 - A).get(), 1 - a, rtol=RT, atol=AT)
    assert np.allclose((2.0 / (A * A + 1)).get(), 2.0 / (a * a + 1), rtol=RT)
    pos = np.abs(a) + 0.5
    assert np.allclose((hip.from_numpy(pos) ** 0.5).get(), pos ** 0.5, rtol=RT)
    assert np.allclose((hip.from_numpy(pos) ** hip.from_numpy(b)).get(), pos ** b, rtol=1e-4)
    assert np.allclose(hip.maximum(0.0, A).get(), np.maximum(0.0, a))
    assert np.allclose(hip.minimum(A, B).get(), np.minimum(a, b))
    for name in ("exp", "abs", "sign"):
        assert np.allclose(getattr(hip, name)(A).get(), getattr(np, name)(a), rtol=RT, atol=AT)
    assert np.allclose(hip.log(hip.from_numpy(pos)).get(), np.log(pos), rtol=RT, atol=AT)
    assert np.allclose((-A).get(), -a)


def test_elementwise_views_inplace_and_cast(hip):
    rng = np.random.default_rng(5)
    a = rng.standard_normal((6, 8), dtype=np.float32)
    A = hip.from_numpy(a)
    at = A.T
    assert np.array_equal(at.get(), a.T)
    assert np.array_equal((at + 1).get(), a.T + 1)
    A2 = hip.from_numpy(a.copy())
    A2[1:3, ::2] = 7.0
    ref = a.copy(); ref[1:3, ::2] = 7.0
    assert np.array_equal(A2.get(), ref)
    A2[:, 0] = hip.from_numpy(np.arange(6, dtype=np.float32))
    ref[:, 0] = np.arange(6)
    assert np.array_equal(A2.get(), ref)
    A2 += hip.from_numpy(a)
    ref += a
    assert np.allclose(A2.get(), ref)
    A2 *= 0.5; ref *= 0.5
    assert np.allclose(A2.get(), ref)
    assert np.array_equal(A.astype(np.float64).get(), a.astype(np.float64))
    assert np.array_equal(A.astype(np.int64).get(), a.astype(np.int64))
    m = a > 0
    A3 = hip.from_numpy(a.copy()); A3[hip.from_numpy(m)] = -1.0
    ref3 = a.copy(); ref3[m] = -1.0
    assert np.array_equal(A3.get(), ref3)
    r = A.reshape(2, 3, 8).transpose(1, 0, 2).reshape(6, 8)  # forces a copy
    assert np.array_equal(r.get(), a.reshape(2, 3, 8).transpose(1, 0, 2).reshape(6, 8))
    assert np.array_equal(hip.concatenate([A, A2], axis=1).get(), np.concatenate([a, ref], axis=1))
    assert np.array_equal(hip.pad(A, [(0, 0), (1, 2)]).get(), np.pad(a, [(0, 0), (1, 2)]))


def test_strided_elementwise_index_walks(hip):
    """The strided / broadcasting kernels walk their index with multiply-high divisions (< 2^31 elements) and, where the
    output's innermost extent is a multiple of four at unit stride, with four elements per lane (csrc/elementwise.hip):
    bit-equal to NumPy over the eligibility edges -- innermost extents 4 / 8 / 12 / odd, stride-2 and stride-0 operands,
    bases off the 16-byte grid, transposed views, sliced outputs, in-place forms, extents that are powers of two / one
    more / one less -- and on the 64-bit path above 2^31 elements' worth of index (a large broadcast)."""
    rng = np.random.default_rng(12)

    def both(shape):
        a = rng.standard_normal(shape, dtype=np.float32)
        return a, hip.from_numpy(a)
    # RoPE's shapes: a stride-2 view times a broadcast table (tests/models_plain_llama.py: rotate_pairs)
    x, X = both((3, 17, 6, 24, 2))
    c, C = both((17, 1, 24))
    for k in (0, 1):
        assert np.array_equal((X[..., k] * C).get(), x[..., k] * c)
        assert np.array_equal((X[..., k] * C - X[..., 1 - k]).get(), x[..., k] * c - x[..., 1 - k])
    for inner in (4, 8, 12, 7, 33, 64, 65, 63):
        a, A = both((5, 3, inner))
        b, B = both((3, 1))
        v, V = both((inner,))
        assert np.array_equal((A + B).get(), a + b)                       # stride-0 innermost operand
        assert np.array_equal((A * V).get(), a * v)                       # row broadcast
        assert np.array_equal((A.transpose(1, 0, 2) - 2.0).get(), a.transpose(1, 0, 2) - 2.0)
        assert np.array_equal(hip.ascontiguousarray(A.transpose(1, 0, 2)).get(), np.ascontiguousarray(a.transpose(1, 0, 2)))
        assert np.allclose(hip.exp(A.transpose(1, 0, 2)).get(), np.exp(a.transpose(1, 0, 2)), rtol=RT)
        if inner > 4:                                                       # operand bases off the 16-byte grid
            assert np.array_equal((A[..., 1:] + A[..., :-1]).get(), a[..., 1:] + a[..., :-1])
            assert np.array_equal((A[:, :, 1:inner - 3] * 3.0).get(), a[:, :, 1:inner - 3] * 3.0)
        O = hip.zeros((5, 3, inner + 4), np.float32)                        # a sliced (row-padded) output
        O[..., :inner] = A.transpose(0, 1, 2) * 1.0
        ref = np.zeros((5, 3, inner + 4), np.float32); ref[..., :inner] = a
        assert np.array_equal(O.get(), ref)
        A2 = hip.from_numpy(a.copy()); A2 += B; A2 *= V                     # in place
        assert np.array_equal(A2.get(), (a + b) * v)
    y, Y = both((4, 16, 6, 48))                                             # the transposed copy in front of the scores
    assert np.array_equal(hip.ascontiguousarray(Y.transpose(0, 2, 1, 3)).get(), np.ascontiguousarray(y.transpose(0, 2, 1, 3)))
    assert np.array_equal(hip.ascontiguousarray(Y.transpose(0, 2, 3, 1)).get(), np.ascontiguousarray(y.transpose(0, 2, 3, 1)))
    for ext in (2, 3, 255, 256, 257, 1023, 1025):                           # divisors around powers of two
        a, A = both((3, ext, 8))
        assert np.array_equal((A.transpose(1, 0, 2) + 1.0).get(), a.transpose(1, 0, 2) + 1.0)
        assert np.array_equal((A.transpose(2, 1, 0) + 1.0).get(), a.transpose(2, 1, 0) + 1.0)
    from pydynet_amd import _lib
    if type(_lib.lib()).__name__ == "EmulatedLib":                          # (8.6 GB of host memory: the device only)
        return
    big = hip.from_numpy(rng.standard_normal((1, 2048), dtype=np.float32))   # 2^31 + 2^21 index values: the 64-bit walk
    col = hip.from_numpy(rng.standard_normal((1024 * 1025, 1), dtype=np.float32))
    out = big + col
    hb, hc = big.get(), col.get()
    for r in (0, 1, 524287, 1024 * 1025 - 1):
        assert np.array_equal(out[r].get(), hb[0] + hc[r, 0])
    del out


@pytest.mark.parametrize("shape,axis,keep", [((7, 288), -1, True), ((4, 6, 256), (0, 1), False),
                                             ((300, 1000), None, False), ((5000, 33), 0, False),
                                             ((3, 4, 5), 1, True), ((2, 3, 4, 5), (1, 3), False),
                                             ((70000,), 0, False)])
def test_reductions(hip, shape, axis, keep):
    rng = np.random.default_rng(6)
    a = rng.standard_normal(shape, dtype=np.float32)
    A = hip.from_numpy(a)
    for name in ("sum", "mean", "max", "min"):
        got = getattr(A, name)(axis, keep).get()
        ref = getattr(a, name)(axis=axis, keepdims=keep)
        assert got.shape == ref.shape, name
        assert np.allclose(got, ref, rtol=1e-4, atol=1e-4), name
    if axis is None or isinstance(axis, int):
        assert np.array_equal(A.argmax(axis).get(), a.argmax(axis))
        assert np.array_equal(A.argmin(axis).get(), a.argmin(axis))


def test_reduce_transposed_view_and_ties(hip):
    a = np.array([[1, 5, 5, 2], [9, 9, 0, 9]], dtype=np.float32)
    A = hip.from_numpy(a)
    assert np.array_equal(A.argmax(1).get(), a.argmax(1))
    assert np.array_equal(A.T.sum(0).get(), a.T.sum(0))
    assert np.array_equal((A == A).sum().get(), (a == a).sum())


# ---------------------------------------------------------------------------------------
def np_softmax(x, axis=-1):
    m = x.max(axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis, keepdims=True)


def test_softmax_fwd_bwd_and_causal(hip):
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(7)
    for rows, cols in [(24, 256), (5, 288), (3, 32000), (4, 10)]:
        x = rng.standard_normal((rows, cols), dtype=np.float32) * 3
        dy = rng.standard_normal((rows, cols), dtype=np.float32)
        X, DY = hip.from_numpy(x), hip.from_numpy(dy)
        Y, DX = hip.empty((rows, cols)), hip.empty((rows, cols))
        L.call("pdn_softmax_fwd_f32", X._ptr, Y._ptr, rows, cols, 1.0, 0, 0, hip.stream())
        y = np_softmax(x)
        assert np.allclose(Y.get(), y, rtol=2e-5, atol=1e-7)
        L.call("pdn_softmax_bwd_f32", Y._ptr, DY._ptr, DX._ptr, rows, cols, 1.0, hip.stream())
        dx = (dy - (dy * y).sum(-1, keepdims=True)) * y
        assert np.allclose(DX.get(), dx, rtol=1e-4, atol=1e-6)
    # attention prologue: / sqrt(hd) + causal mask, rows = (b*h, L)
    Lq, hd = 64, 48
    s = rng.standard_normal((3, Lq, Lq), dtype=np.float32)
    mask = np.triu(np.full((Lq, Lq), -np.inf, dtype=np.float32), 1)
    ref = np_softmax(s / np.float32(math.sqrt(hd)) + mask)
    S, P = hip.from_numpy(s), hip.empty(s.shape)
    L.call("pdn_softmax_fwd_f32", S._ptr, P._ptr, 3 * Lq, Lq, math.sqrt(hd), Lq, 0, hip.stream())
    got = P.get()
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-7)
    assert np.all(got[:, np.triu_indices(Lq, 1)[0], np.triu_indices(Lq, 1)[1]] == 0.0)


def test_rmsnorm_fwd_bwd(hip):
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(8)
    rows, cols, eps = 700, 288, 1e-6
    x = rng.standard_normal((rows, cols), dtype=np.float32)
    w = rng.standard_normal(cols, dtype=np.float32)
    dy = rng.standard_normal((rows, cols), dtype=np.float32)
    X, W, DY = map(hip.from_numpy, (x, w, dy))
    Y, R, DX = hip.empty((rows, cols)), hip.empty((rows,)), hip.empty((rows, cols))
    DW = hip.zeros((cols,), np.float32)
    L.call("pdn_rmsnorm_fwd_f32", X._ptr, W._ptr, Y._ptr, R._ptr, rows, cols, eps, hip.stream())
    r = np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + eps)
    z = x / r
    assert np.allclose(Y.get(), z * w, rtol=2e-5, atol=2e-6)
    ws, wsb = hip.workspace(L.query("pdn_rmsnorm_bwd_workspace_bytes", rows, cols))
    L.call("pdn_rmsnorm_bwd_f32", X._ptr, W._ptr, R._ptr, DY._ptr, None, DX._ptr, DW._ptr, 1, rows, cols,
           ws, wsb, hip.stream())
    dz = dy * w
    dx = (dz - z * (z * dz).mean(-1, keepdims=True)) / r
    assert np.allclose(DX.get(), dx, rtol=1e-4, atol=1e-5)
    assert np.allclose(DW.get(), (dy * z).sum(0), rtol=1e-4, atol=1e-4)
    # gradient already held by x folded into the same pass; dw skipped
    res = rng.standard_normal((rows, cols), dtype=np.float32)
    RES, DX2 = hip.from_numpy(res), hip.empty((rows, cols))
    L.call("pdn_rmsnorm_bwd_f32", X._ptr, W._ptr, R._ptr, DY._ptr, RES._ptr, DX2._ptr, None, 0, rows,
           cols, None, 0, hip.stream())
    assert np.allclose(DX2.get(), dx + res, rtol=1e-4, atol=1e-5)


def test_swiglu_rope_relu(hip):
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(9)
    n = 1000 * 768 + 3
    g = rng.standard_normal(n, dtype=np.float32) * 2
    u = rng.standard_normal(n, dtype=np.float32)
    dy = rng.standard_normal(n, dtype=np.float32)
    G, U, DY = map(hip.from_numpy, (g, u, dy))
    Y, DG, DU = hip.empty((n,)), hip.empty((n,)), hip.empty((n,))
    L.call("pdn_swiglu_fwd_f32", G._ptr, U._ptr, Y._ptr, n, hip.stream())
    g64 = g.astype(np.float64)
    s = 1 / (1 + np.exp(-g64))
    assert np.allclose(Y.get(), g64 * s * u, rtol=2e-5, atol=2e-6)
    L.call("pdn_swiglu_bwd_f32", G._ptr, U._ptr, DY._ptr, DG._ptr, DU._ptr, n, hip.stream())
    assert np.allclose(DU.get(), dy * g64 * s, rtol=2e-5, atol=2e-6)
    assert np.allclose(DG.get(), dy * u * s * (1 + g64 * (1 - s)), rtol=1e-4, atol=1e-5)
    L.call("pdn_swiglu_fwd_f32", G._ptr, None, Y._ptr, n, hip.stream())
    assert np.allclose(Y.get(), g64 * s, rtol=2e-5, atol=2e-6)
    # relu backward: gradient 1 at x == 0 (reference quirk)
    x = np.array([-1.0, 0.0, 2.0, -0.0], dtype=np.float32)
    X, D, ONES = hip.from_numpy(x), hip.empty((4,)), hip.from_numpy(np.ones(4, np.float32))
    L.call("pdn_relu_bwd_f32", X._ptr, ONES._ptr, D._ptr, 4, hip.stream())
    assert np.array_equal(D.get(), np.array([0, 1, 1, 1], np.float32))
    # RoPE
    B, Lq, H, hd = 2, 16, 6, 48
    xq = rng.standard_normal((B, Lq, H, hd), dtype=np.float32)
    inv = 1.0 / (10000 ** (np.arange(0, hd, 2) / hd))
    fr = np.outer(np.arange(Lq), inv).astype(np.float32)
    cos, sin = np.cos(fr), np.sin(fr)
    xr, xi = xq[..., 0::2], xq[..., 1::2]
    c, s_ = cos[None, :, None, :], sin[None, :, None, :]
    ref = np.stack([xr * c - xi * s_, xr * s_ + xi * c], -1).reshape(xq.shape)
    XQ, OUT, COS, SIN = hip.from_numpy(xq), hip.empty(xq.shape), hip.from_numpy(cos), hip.from_numpy(sin)
    L.call("pdn_rope_f32", XQ._ptr, COS._ptr, SIN._ptr, OUT._ptr,
           B * Lq, Lq, H, hd, 0, hip.stream())
    assert np.allclose(OUT.get(), ref, rtol=2e-5, atol=2e-6)
    BACK = hip.empty(xq.shape)
    L.call("pdn_rope_f32", OUT._ptr, COS._ptr, SIN._ptr, BACK._ptr,
           B * Lq, Lq, H, hd, 1, hip.stream())
    assert np.allclose(BACK.get(), xq, rtol=1e-4, atol=1e-5)   # rotation by -theta inverts it


def test_embedding_gather_scatter_assign_bit_exact(hip):
    rng = np.random.default_rng(10)
    V, D = 500, 288
    w = rng.standard_normal((V, D), dtype=np.float32)
    ids = rng.integers(0, V, size=(4, 64))
    ids[0, :5] = 7          # duplicates: last write must win
    W = hip.from_numpy(w)
    out = W[hip.from_numpy(ids)]
    assert np.array_equal(out.get(), w[ids])
    assert np.array_equal(W[ids.tolist()].get(), w[ids])
    g = rng.standard_normal((4, 64, D), dtype=np.float32)
    full = hip.zeros((V, D), np.float32)
    full[hip.from_numpy(ids)] = hip.from_numpy(g)
    ref = np.zeros((V, D), np.float32); ref[ids] = g
    assert np.array_equal(full.get(), ref)
    # pick one column per row and its scatter
    x = rng.standard_normal((50, 33), dtype=np.float32)
    t = rng.integers(0, 33, size=50)
    assert np.array_equal(hip.from_numpy(x)[range(50), hip.from_numpy(t)].get(), x[range(50), t])
    z = hip.zeros((50, 33), np.float32)
    gv = rng.standard_normal(50, dtype=np.float32)
    z[range(50), hip.from_numpy(t)] = hip.from_numpy(gv)
    zr = np.zeros((50, 33), np.float32); zr[range(50), t] = gv
    assert np.array_equal(z.get(), zr)
    hip.check_index_errors()


def test_cross_entropy_fused(hip):
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(11)
    for rows, V in [(64, 32000), (37, 10)]:
        x = rng.standard_normal((rows, V), dtype=np.float32) * 2
        t = rng.integers(0, V, size=rows)
        X, T = hip.from_numpy(x), hip.from_numpy(t)
        lr, lse, out = hip.empty((rows,)), hip.empty((rows,)), hip.empty((1,))
        err = hip.zeros((1,), np.int32)
        L.call("pdn_cross_entropy_fwd_f32", X._ptr, T._ptr, rows, V, 1, lr._ptr, lse._ptr, out._ptr,
               err._ptr, hip.stream())
        x64 = x.astype(np.float64)
        m = x64.max()
        l = np.log(np.exp(x64 - m).sum(1)) + m
        ref = (l - x64[np.arange(rows), t]).mean()
        assert abs(out.get()[0] - ref) < 1e-5 * abs(ref)
        DX = hip.empty((rows, V))
        L.call("pdn_cross_entropy_bwd_f32", X._ptr, T._ptr, lse._ptr, None, 1.0 / rows, DX._ptr, rows, V, hip.stream())
        sm = np.exp(x64 - l[:, None]); sm[np.arange(rows), t] -= 1
        assert np.allclose(DX.get(), sm / rows, rtol=1e-4, atol=1e-8)
        assert int(err.get()[0]) == 0


def test_adam_multi_matches_reference_update(hip):
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(12)
    sizes = [288 * 288, 288, 1000, 7]
    P = [rng.standard_normal(n, dtype=np.float32) for n in sizes]
    G = [rng.standard_normal(n, dtype=np.float32) for n in sizes]
    M = [np.zeros(n, np.float32) for n in sizes]
    Vv = [np.zeros(n, np.float32) for n in sizes]
    dP, dG, dM, dV = [[hip.from_numpy(a) for a in lst] for lst in (P, G, M, Vv)]
    CH = 16384
    rows = []
    for p, g, m, v in zip(dP, dG, dM, dV):
        for off in range(0, p.size, CH):
            n = min(CH, p.size - off)
            rows.append([p._ptr + 4 * off, g._ptr + 4 * off, m._ptr + 4 * off, v._ptr + 4 * off, n])
    table = hip.from_numpy(np.asarray(rows, dtype=np.int64))
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    for t in (1, 2, 3):
        a_t = math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        L.call("pdn_adam_multi_f32", table._ptr, len(rows), lr * a_t, b1, b2, 1 - b1, 1 - b2,
               eps, 0.0, 1.0, hip.stream())
        for i in range(len(sizes)):      # optimizer.py:187-195, verbatim order of operations
            grad = G[i]
            M[i] *= b1; M[i] += (1 - b1) * grad
            Vv[i] *= b2; Vv[i] += (1 - b2) * grad ** 2
            P[i] -= lr * a_t * M[i] / (Vv[i] ** 0.5 + eps)
    for i in range(len(sizes)):
        assert np.allclose(dP[i].get(), P[i], rtol=1e-5, atol=1e-7)
        assert np.allclose(dM[i].get(), M[i], rtol=1e-5, atol=1e-8)
        assert np.allclose(dV[i].get(), Vv[i], rtol=1e-5, atol=1e-10)


def test_gemm_residual_and_fused_bias_gradient(hip):
    rng = np.random.default_rng(13)
    # residual add in the epilogue (interior tiles and ragged edges)
    for M, N, K in [(256, 288, 768), (100, 52, 36)]:
        a = rng.standard_normal((M, K), dtype=np.float32)
        b = rng.standard_normal((K, N), dtype=np.float32)
        r = rng.standard_normal((M, N), dtype=np.float32)
        C = hip.empty((M, N))
        hip.gemm(hip.from_numpy(a), hip.from_numpy(b), C, residual=hip.from_numpy(r))
        assert rel_err(C.get(), a.astype(np.float64) @ b + r) < 1e-5
    # dW = x^T @ g with the column sums of g (bias gradient) produced by the same kernel
    for T, fin, fout in [(2048, 288, 32000), (512, 96, 200), (8192, 288, 288)]:
        x = rng.standard_normal((T, fin), dtype=np.float32)
        g = rng.standard_normal((T, fout), dtype=np.float32)
        dw0 = rng.standard_normal((fin, fout), dtype=np.float32)
        db0 = rng.standard_normal(fout, dtype=np.float32)
        DW, DB = hip.from_numpy(dw0.copy()), hip.from_numpy(db0.copy())
        hip.gemm(hip.from_numpy(x).T, hip.from_numpy(g), DW, beta=1.0, b_colsum=DB, colsum_accumulate=True)
        assert rel_err(DW.get(), dw0 + x.T.astype(np.float64) @ g) < 2e-5
        assert rel_err(DB.get(), db0 + g.astype(np.float64).sum(0)) < 2e-5


def test_cross_entropy_one_pass_forward_backward(hip):
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(14)
    for rows, V in [(48, 32000), (19, 10), (700, 32000)]:
        x = rng.standard_normal((rows, V), dtype=np.float32) * 2
        t = rng.integers(0, V, size=rows)
        X, T = hip.from_numpy(x), hip.from_numpy(t)
        lr, lse, out, DX = hip.empty((rows,)), hip.empty((rows,)), hip.empty((1,)), hip.empty((rows, V))
        wsb = L.query("pdn_cross_entropy_colsum_workspace_bytes", rows, V)
        assert (wsb > 0) == (V == 32000)
        CS = hip.empty((V,)) if wsb else None
        ws, wsb = hip.workspace(wsb) if wsb else (None, 0)
        L.call("pdn_cross_entropy_fwd_bwd_f32", X._ptr, T._ptr, rows, V, 1, 1.0 / rows, lr._ptr, lse._ptr,
               out._ptr, DX._ptr, CS._ptr if CS is not None else None, ws, wsb,
               hip.err_flag_ptr(), hip.stream())
        x64 = x.astype(np.float64)
        l = np.log(np.exp(x64 - x64.max()).sum(1)) + x64.max()
        assert abs(out.get()[0] - (l - x64[np.arange(rows), t]).mean()) < 1e-5 * abs(l.mean())
        sm = np.exp(x64 - l[:, None]); sm[np.arange(rows), t] -= 1
        assert np.allclose(DX.get(), sm / rows, rtol=1e-4, atol=1e-8)
        if CS is not None:
            assert np.allclose(CS.get(), (sm / rows).sum(0), rtol=1e-4, atol=1e-7)
        one, half = hip.from_numpy(np.ones(1, np.float32)), hip.from_numpy(np.full(1, 0.5, np.float32))
        before = DX.get()
        L.call("pdn_scale_by_device_scalar_f32", DX._ptr, DX.size, one._ptr, hip.stream())
        assert np.array_equal(DX.get(), before)                       # scalar 1: buffer untouched
        L.call("pdn_scale_by_device_scalar_f32", DX._ptr, DX.size, half._ptr, hip.stream())
        assert np.array_equal(DX.get(), before * np.float32(0.5))
    hip.check_index_errors()


@pytest.mark.parametrize("B,H,L,causal,hd", [(2, 6, 256, 1, 48), (3, 2, 64, 1, 48), (1, 6, 128, 0, 48), (2, 3, 32, 1, 48),
                                             (2, 2, 512, 1, 48),      # two 256-key chunks, one online rescale
                                             (1, 2, 1024, 1, 48),     # max_seq_len of llm/llama/finetune.py:44
                                             (1, 2, 512, 0, 48),      # not causal: every query group visits every chunk
                                             (1, 3, 736, 1, 48),      # ragged last chunk (23 tiles: 8 + 8 + 7)
                                             (2, 4, 256, 1, 64), (1, 2, 96, 0, 64), (1, 2, 640, 1, 64),    # head dim 64
                                             # more heads than CUs: the persistent kernels of csrc/attention_p.hip walk
                                             # several heads per workgroup (double-buffered images, prefetched operands)
                                             (101, 6, 64, 1, 48), (43, 7, 96, 0, 48), (3, 100, 256, 1, 48)])
def test_fused_attention_forward_backward(hip, B, H, L, causal, hd):
    """Fused attention vs a float64 statement of llm/llama/model.py:112-121 and its gradients."""
    from pydynet_amd import _lib
    Lb = _lib.lib()
    rng = np.random.default_rng(100 + L)
    q, k, v, do = (rng.standard_normal((B, L, H, hd), dtype=np.float32) for _ in range(4))
    k[0, L // 2, 0] *= 6.0                       # a spiky key: large scores exercise the max shift
    dq, dk, dv, o = (hip.empty((B, L, H, hd)) for _ in range(4))
    Q, K, V, DO = map(hip.from_numpy, (q, k, v, do))
    lse = hip.empty((B, H, L))
    Lb.call("pdn_attention_fwd_f32", Q._ptr, K._ptr, V._ptr, o._ptr, lse._ptr, B, H, L, hd, H * hd, L * H * hd,
            H * hd, L * H * hd, causal, None, None, hip.stream())
    q64, k64, v64, g64 = (a.astype(np.float64).transpose(0, 2, 1, 3) for a in (q, k, v, do))
    s = q64 @ k64.swapaxes(-1, -2) / math.sqrt(hd)
    if causal:
        s = s + np.triu(np.full((L, L), -np.inf), 1)
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    p = e / e.sum(-1, keepdims=True)
    ref_o = (p @ v64).transpose(0, 2, 1, 3)
    assert rel_err(o.get(), ref_o) < 2e-5
    assert np.allclose(lse.get(), (m + np.log(e.sum(-1, keepdims=True)))[..., 0], rtol=1e-5, atol=1e-5)
    ws, wsb = hip.workspace(Lb.query("pdn_attention_bwd_workspace_bytes", B, H, L))
    Lb.call("pdn_attention_bwd_f32", Q._ptr, K._ptr, V._ptr, o._ptr, DO._ptr, lse._ptr, dq._ptr, dk._ptr, dv._ptr,
            B, H, L, hd, H * hd, L * H * hd, H * hd, L * H * hd, causal, None, None, ws, wsb, hip.stream())
    dp = g64 @ v64.swapaxes(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / math.sqrt(hd)
    assert rel_err(dv.get(), (p.swapaxes(-1, -2) @ g64).transpose(0, 2, 1, 3)) < 5e-5
    assert rel_err(dq.get(), (ds @ k64).transpose(0, 2, 1, 3)) < 5e-5
    assert rel_err(dk.get(), (ds.swapaxes(-1, -2) @ q64).transpose(0, 2, 1, 3)) < 5e-5
    if causal:                                    # masked probabilities are exactly zero: the first
        assert np.allclose(o.get()[:, 0], v[:, 0], rtol=1e-6)   # query attends only to key 0


@pytest.mark.parametrize("B,H,L,causal,hd,shared", [(3, 4, 64, 0, 48, False), (2, 2, 256, 1, 48, False), (2, 3, 96, 0, 64, True),
                                                    (1, 2, 512, 0, 48, False),     # two key chunks, online rescale
                                                    (2, 2, 640, 1, 64, False), (70, 6, 64, 0, 48, False)])
def test_fused_attention_key_bias(hip, B, H, L, causal, hd, shared):
    """The resident kernels with an additive key bias -- the (B, 1, 1, L) padding mask of examples/pydynet/transformer.py:92-96
    (-inf on masked keys) plus finite entries -- vs a float64 statement of `softmax(q k^T / sqrt(hd) + mask) v`."""
    from pydynet_amd import _lib
    Lb = _lib.lib()
    rng = np.random.default_rng(300 + L + hd)
    q, k, v, do = (rng.standard_normal((B, L, H, hd), dtype=np.float32) for _ in range(4))
    nb = 1 if shared else B
    kb = (0.5 * rng.standard_normal((nb, L))).astype(np.float32)
    for b in range(nb):                            # ragged valid lengths: the tail of every row is masked out
        kb[b, L - 5 - (9 * b) % (L // 2):] = -np.inf
        kb[b, 3 + b % 8] = -np.inf                     # and one key in the middle (never key 0: causal rows stay non-empty)
    Q, K, V, DO, KB = map(hip.from_numpy, (q, k, v, do, kb))
    o, dq, dk, dv = (hip.empty((B, L, H, hd)) for _ in range(4))
    lse = hip.empty((B, H, L))
    kbs = 0 if shared else L
    Lb.call("pdn_attention_fwd_bias_f32", Q._ptr, K._ptr, V._ptr, o._ptr, lse._ptr, B, H, L, hd, H * hd, L * H * hd,
            H * hd, L * H * hd, causal, KB._ptr, kbs, hip.stream())
    q64, k64, v64, g64 = (a.astype(np.float64).transpose(0, 2, 1, 3) for a in (q, k, v, do))
    s = q64 @ k64.swapaxes(-1, -2) / math.sqrt(hd) + np.broadcast_to(kb.astype(np.float64)[:, None, None, :], (nb, 1, 1, L))
    if causal:
        s = s + np.triu(np.full((L, L), -np.inf), 1)
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    p = e / e.sum(-1, keepdims=True)
    assert rel_err(o.get(), (p @ v64).transpose(0, 2, 1, 3)) < 2e-5
    assert np.allclose(lse.get(), (m + np.log(e.sum(-1, keepdims=True)))[..., 0], rtol=1e-5, atol=1e-5)
    ws, wsb = hip.workspace(Lb.query("pdn_attention_bwd_workspace_bytes", B, H, L))
    Lb.call("pdn_attention_bwd_bias_f32", Q._ptr, K._ptr, V._ptr, o._ptr, DO._ptr, lse._ptr, dq._ptr, dk._ptr, dv._ptr,
            B, H, L, hd, H * hd, L * H * hd, H * hd, L * H * hd, causal, KB._ptr, kbs, ws, wsb, hip.stream())
    dp = g64 @ v64.swapaxes(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / math.sqrt(hd)
    assert rel_err(dv.get(), (p.swapaxes(-1, -2) @ g64).transpose(0, 2, 1, 3)) < 5e-5
    assert rel_err(dq.get(), (ds @ k64).transpose(0, 2, 1, 3)) < 5e-5
    assert rel_err(dk.get(), (ds.swapaxes(-1, -2) @ q64).transpose(0, 2, 1, 3)) < 5e-5
    masked = ~np.isfinite(kb)
    got_dk, got_dv = dk.get(), dv.get()
    for b in range(B):                             # a masked key receives no gradient at all
        assert not got_dk[b][masked[b if not shared else 0]].any() and not got_dv[b][masked[b if not shared else 0]].any()


@pytest.mark.parametrize("L,hd,pad", [(512, 48, 256), (768, 64, 300), (1024, 48, 520)])
def test_fused_attention_left_padded_batch_is_finite(hip, L, hd, pad):
    """A LEFT-padded batch: the first `pad` >= 256 keys of a row carry -inf, i.e. the whole first 256-key chunk of the
    chunked resident kernel is masked and its running maximum is still -inf when the first finite score arrives (ADVICE
    round 4: exp2(-inf * c + inf) = NaN poisoned the row).  The reference -- softmax over the full row,
    nn/functional.py:43-49 with the (B, 1, 1, L) mask of examples/pydynet/transformer.py:92-96 -- is finite."""
    from pydynet_amd import _lib
    Lb = _lib.lib()
    B, H = 2, 2
    rng = np.random.default_rng(L + pad)
    q, k, v, do = (rng.standard_normal((B, L, H, hd), dtype=np.float32) for _ in range(4))
    kb = np.zeros((B, L), np.float32)
    kb[0, :pad] = -np.inf                          # sequence 0: left padding over more than one chunk
    kb[1, :7] = -np.inf                            # sequence 1: a few pad tokens only
    Q, K, V, DO, KB = map(hip.from_numpy, (q, k, v, do, kb))
    o, dq, dk, dv = (hip.empty((B, L, H, hd)) for _ in range(4))
    lse = hip.empty((B, H, L))
    Lb.call("pdn_attention_fwd_bias_f32", Q._ptr, K._ptr, V._ptr, o._ptr, lse._ptr, B, H, L, hd, H * hd, L * H * hd,
            H * hd, L * H * hd, 0, KB._ptr, L, hip.stream())
    q64, k64, v64, g64 = (a.astype(np.float64).transpose(0, 2, 1, 3) for a in (q, k, v, do))
    s = q64 @ k64.swapaxes(-1, -2) / math.sqrt(hd) + kb.astype(np.float64)[:, None, None, :]
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    p = e / e.sum(-1, keepdims=True)
    assert np.isfinite(o.get()).all() and np.isfinite(lse.get()).all()
    assert rel_err(o.get(), (p @ v64).transpose(0, 2, 1, 3)) < 2e-5
    assert np.allclose(lse.get(), (m + np.log(e.sum(-1, keepdims=True)))[..., 0], rtol=1e-5, atol=1e-5)
    ws, wsb = hip.workspace(Lb.query("pdn_attention_bwd_workspace_bytes", B, H, L))
    Lb.call("pdn_attention_bwd_bias_f32", Q._ptr, K._ptr, V._ptr, o._ptr, DO._ptr, lse._ptr, dq._ptr, dk._ptr, dv._ptr,
            B, H, L, hd, H * hd, L * H * hd, H * hd, L * H * hd, 0, KB._ptr, L, ws, wsb, hip.stream())
    dp = g64 @ v64.swapaxes(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / math.sqrt(hd)
    for got, ref in ((dv, p.swapaxes(-1, -2) @ g64), (dq, ds @ k64), (dk, ds.swapaxes(-1, -2) @ q64)):
        assert np.isfinite(got.get()).all()
        assert rel_err(got.get(), ref.transpose(0, 2, 1, 3)) < 5e-5
    assert not dk.get()[0, :pad].any() and not dv.get()[0, :pad].any()       # padded keys receive no gradient


@pytest.mark.parametrize("B,H,L,causal,hd,with_mask,kind", [(2, 2, 50, 0, 64, False, "stream"),   # CLIP vision: 49 patches + class token
                                                            (2, 2, 77, 1, 64, False, "stream"),   # CLIP text
                                                            (3, 4, 64, 0, 48, True, "resident"),
                                                            (2, 3, 128, 1, 64, True, "resident"),
                                                            (2, 2, 40, 0, 48, True, "stream")])
def test_attention_node_masks_and_ragged_lengths(hip, B, H, L, causal, hd, with_mask, kind):
    """`fused.attention` with a (B, 1, 1, L) padding mask and / or a length that is not a multiple of 32: key-only masks
    on whole tiles take the resident kernels (key bias), everything else the streaming ones; both agree with float64."""
    import pydynet_amd as pdn
    from pydynet_amd.core import fused
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    rng = np.random.default_rng(17 * L + hd)
    q, k, v, w = (rng.standard_normal((B, L, H, hd), dtype=np.float32) for _ in range(4))
    mask = None
    if with_mask:
        mask = np.zeros((B, 1, 1, L), np.float32)
        for b in range(B):
            mask[b, 0, 0, L - 3 - 4 * b:] = -np.inf
    tq, tk, tv = (pdn.Tensor(a, dtype=np.float32, device="hip:0", requires_grad=True) for a in (q, k, v))
    node = fused.attention(tq, tk, tv, causal=bool(causal),
                           mask=pdn.Tensor(mask, dtype=np.float32, device="hip:0") if with_mask else None)
    assert node._kind == kind, node._kind
    (node * pdn.Tensor(w, dtype=np.float32, device="hip:0")).sum().backward()
    q64, k64, v64, g64 = (a.astype(np.float64).transpose(0, 2, 1, 3) for a in (q, k, v, w))
    s = q64 @ k64.swapaxes(-1, -2) / math.sqrt(hd)
    if with_mask:
        s = s + mask.astype(np.float64)
    if causal:
        s = s + np.triu(np.full((L, L), -np.inf), 1)
    e = np.exp(s - s.max(-1, keepdims=True))
    p = e / e.sum(-1, keepdims=True)
    assert rel_err(node.numpy(), (p @ v64).transpose(0, 2, 1, 3)) < 2e-5
    dp = g64 @ v64.swapaxes(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / math.sqrt(hd)
    assert rel_err(tv.grad.get(), (p.swapaxes(-1, -2) @ g64).transpose(0, 2, 1, 3)) < 5e-5
    assert rel_err(tq.grad.get(), (ds @ k64).transpose(0, 2, 1, 3)) < 5e-5
    assert rel_err(tk.grad.get(), (ds.swapaxes(-1, -2) @ q64).transpose(0, 2, 1, 3)) < 5e-5


def test_colnorm_forward_backward_large_offset(hip):
    """Reference-LayerNorm kernels on the transformer example's shape, with a large common offset
    (a one-pass E[x^2]-E[x]^2 variance would lose every digit here)."""
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(21)
    rows, cols, eps, mom = 5632 + 37, 512, 1e-6, 0.1
    x = (rng.standard_normal((rows, cols)) * 0.5 + 300.0).astype(np.float32)
    w, b = rng.standard_normal(cols).astype(np.float32), rng.standard_normal(cols).astype(np.float32)
    dy = rng.standard_normal((rows, cols)).astype(np.float32)
    rm0, rv0 = rng.standard_normal(cols).astype(np.float32), rng.random(cols).astype(np.float32)
    X, W, Bb, DY = map(hip.from_numpy, (x, w, b, dy))
    RM, RV = hip.from_numpy(rm0.copy()), hip.from_numpy(rv0.copy())
    Y, MU, RS = hip.empty((rows, cols)), hip.empty((cols,)), hip.empty((cols,))
    ws, wsb = hip.workspace(L.query("pdn_colnorm_workspace_bytes", rows, cols))
    L.call("pdn_colnorm_fwd_f32", X._ptr, W._ptr, Bb._ptr, Y._ptr, MU._ptr, RS._ptr, RM._ptr, RV._ptr, mom, eps,
           rows, cols, ws, wsb, hip.stream())
    x64 = x.astype(np.float64)
    mu = x64.mean(0); var = ((x64 - mu) ** 2).mean(0); rs = 1 / np.sqrt(var + eps); xh = (x64 - mu) * rs
    assert np.allclose(MU.get(), mu, rtol=1e-6)
    assert np.allclose(RS.get(), rs, rtol=2e-4)          # var of (x - 300) in fp32: ~1e-4 relative
    assert np.allclose(Y.get(), xh * w + b, rtol=1e-3, atol=2e-3)
    assert np.allclose(RM.get(), rm0 * (1 - mom) + mom * mu, rtol=1e-5, atol=1e-6)
    assert np.allclose(RV.get(), rv0 * (1 - mom) + mom * var, rtol=2e-4)
    DX, DW, DB = hip.empty((rows, cols)), hip.from_numpy(np.ones(cols, np.float32)), hip.zeros((cols,), np.float32)
    L.call("pdn_colnorm_bwd_f32", X._ptr, W._ptr, MU._ptr, RS._ptr, DY._ptr, DX._ptr, DW._ptr, DB._ptr, 1, rows,
           cols, ws, wsb, hip.stream())
    g = dy.astype(np.float64)
    xh32 = (x64 - MU.get()) * RS.get()                    # the statistics the kernel actually used
    sdb, sdw = g.sum(0), (g * xh32).sum(0)
    assert np.allclose(DB.get(), sdb, rtol=1e-4, atol=1e-3)
    assert np.allclose(DW.get(), 1 + sdw, rtol=1e-4, atol=2e-3)
    dx = w * RS.get() * (g - sdb / rows - xh32 * (sdw / rows))
    assert np.allclose(DX.get(), dx, rtol=1e-3, atol=1e-4)


def test_gru_gate_kernels(hip):
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(22)
    B, H = 1568, 32
    g1 = (rng.standard_normal((B, 2 * H)) * 3).astype(np.float32)
    g2 = (rng.standard_normal((B, H)) * 3).astype(np.float32)
    h = rng.standard_normal((B, H)).astype(np.float32)
    dhn, drh = rng.standard_normal((B, H)).astype(np.float32), rng.standard_normal((B, H)).astype(np.float32)
    G1, G2, Hh, DHN, DRH = map(hip.from_numpy, (g1, g2, h, dhn, drh))
    Z, R, RH, N, HN = (hip.empty((B, H)) for _ in range(5))
    L.call("pdn_gru_gates_fwd_f32", G1._ptr, Hh._ptr, Z._ptr, R._ptr, RH._ptr, B, H, hip.stream())
    L.call("pdn_gru_out_fwd_f32", G2._ptr, Z._ptr, Hh._ptr, N._ptr, HN._ptr, B, H, hip.stream())
    sig = lambda t: 1 / (1 + np.exp(-t.astype(np.float64)))
    z, r, n = sig(g1[:, :H]), sig(g1[:, H:]), np.tanh(g2.astype(np.float64))
    assert np.allclose(Z.get(), z, rtol=1e-5, atol=1e-7) and np.allclose(R.get(), r, rtol=1e-5, atol=1e-7)
    assert np.allclose(RH.get(), r * h, rtol=1e-5, atol=1e-6)
    assert np.allclose(N.get(), n, rtol=1e-5, atol=1e-6)
    assert np.allclose(HN.get(), (1 - z) * h + z * n, rtol=1e-5, atol=1e-6)
    DG2, DG1, DH = hip.empty((B, H)), hip.empty((B, 2 * H)), hip.empty((B, H))
    L.call("pdn_gru_out_bwd_f32", DHN._ptr, Z._ptr, N._ptr, Hh._ptr, DG2._ptr, DG1._ptr, DH._ptr, B, H, hip.stream())
    L.call("pdn_gru_gates_bwd_f32", DRH._ptr, R._ptr, Hh._ptr, DG1._ptr, DH._ptr, B, H, hip.stream())
    assert np.allclose(DG2.get(), dhn * z * (1 - n * n), rtol=1e-4, atol=1e-6)
    assert np.allclose(DG1.get()[:, :H], dhn * (n - h) * z * (1 - z), rtol=1e-4, atol=1e-6)
    assert np.allclose(DG1.get()[:, H:], drh * h * r * (1 - r), rtol=1e-4, atol=1e-6)
    assert np.allclose(DH.get(), dhn * (1 - z) + drh * r, rtol=1e-4, atol=1e-6)


def test_gemm_batched_weight_gradients_one_launch(hip):
    # three x^T @ d_i products sharing x, written into equally spaced (descending) leaf buffers with
    # beta = 1: the batched form of the wave-streaming kernel used by the fused QKV node
    rng = np.random.default_rng(31)
    T_, D = 4096 + 24, 96
    x = rng.standard_normal((T_, D), dtype=np.float32)
    d = rng.standard_normal((3, T_, D), dtype=np.float32)
    g0 = rng.standard_normal((3, D, D), dtype=np.float32)
    X, Dq = hip.from_numpy(x), hip.from_numpy(d)
    flat = hip.from_numpy(g0.reshape(-1).copy())
    views = [flat[(2 - i) * D * D:(3 - i) * D * D].reshape(D, D) for i in range(3)]   # descending addresses
    stack = hip.stacked_view(views)
    assert stack is not None and stack._strides[0] == -D * D
    hip.gemm(X.T, Dq, stack, beta=1.0)
    for i in range(3):
        ref = g0[2 - i] + x.T.astype(np.float64) @ d[i].astype(np.float64)
        assert rel_err(views[i].get(), ref) < 2e-5
    assert hip.stacked_view([views[0], views[2]]) is not None and hip.stacked_view([views[0], views[1][:, :4]]) is None


@pytest.mark.parametrize("M", [1, 2, 3, 4])
def test_gemm_skinny_rows(hip, M):
    rng = np.random.default_rng(40 + M)
    K, N = 288, 32000
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((K, N), dtype=np.float32)
    bias = rng.standard_normal(N, dtype=np.float32)
    c0 = rng.standard_normal((M, N), dtype=np.float32)
    A, W, Bv = hip.from_numpy(a), hip.from_numpy(w), hip.from_numpy(bias)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    C = hip.from_numpy(c0.copy())
    hip.gemm(A, W, C, beta=1.0, bias=Bv)
    assert rel_err(C.get(), ref + bias + c0) < 1e-5
    C2 = hip.empty((M, N))
    hip.gemm(A, W, C2, alpha=2.0, residual=hip.from_numpy(c0))
    assert rel_err(C2.get(), 2 * ref + c0) < 1e-5


@pytest.mark.parametrize("T_", [1, 7, 300, 1024])
def test_attention_decode_kernel(hip, T_):
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(50 + T_)
    B, H, hd, maxL = 2, 6, 48, 1024
    q = rng.standard_normal((B, H, hd), dtype=np.float32)
    kc = rng.standard_normal((B, maxL, H, hd), dtype=np.float32)
    vc = rng.standard_normal((B, maxL, H, hd), dtype=np.float32)
    Q, KC, VC, O = hip.from_numpy(q), hip.from_numpy(kc), hip.from_numpy(vc), hip.empty((B, H, hd))
    L.call("pdn_attention_decode_f32", Q._ptr, KC._ptr, VC._ptr, O._ptr, B, H, T_, hd, maxL * H * hd, hip.stream())
    s = np.einsum("bhd,bthd->bht", q.astype(np.float64), kc[:, :T_].astype(np.float64)) / math.sqrt(hd)
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    ref = np.einsum("bht,bthd->bhd", p, vc[:, :T_].astype(np.float64))
    assert rel_err(O.get(), ref) < 1e-5


def test_persistent_gru_sequence_equals_per_step_path(hip):
    """pdn_gru_seq_* (the whole time loop in one launch, h in MFMA accumulator registers) against the
    per-step launches of the same node, forward and every gradient; ragged batch (not a multiple of 32)."""
    import pydynet_amd as pdn
    from pydynet_amd.core import fused
    from pydynet_amd.core.tensor import Graph
    rng = np.random.default_rng(4)
    T, B, I, H = 9, 70, 3, 32
    arrs = dict(x=rng.standard_normal((T, B, I)), h0=rng.standard_normal((B, H)),
                wx1=0.3 * rng.standard_normal((I, 2 * H)), wh1=0.3 * rng.standard_normal((H, 2 * H)),
                wx2=0.3 * rng.standard_normal((I, H)), wh2=0.3 * rng.standard_normal((H, H)),
                b1=0.1 * rng.standard_normal((2 * H,)), b2=0.1 * rng.standard_normal((H,)))
    w = rng.standard_normal((T, B, H)).astype(np.float32)
    res = []
    for persistent in (True, False):
        Graph.clear()
        fused.gru_sequence.use_persistent = persistent
        try:
            ts = {k: pdn.Tensor(v.astype(np.float32), dtype=np.float32, device="hip:0", requires_grad=True)
                  for k, v in arrs.items()}
            node = fused.gru_sequence(ts["x"], ts["h0"], ts["wx1"], ts["wh1"], ts["wx2"], ts["wh2"], ts["b1"], ts["b2"])
            assert node._persistent == persistent
            (node * pdn.Tensor(w, dtype=np.float32, device="hip:0")).sum().backward()
            res.append([node.numpy()] + [ts[k].grad.get() for k in arrs])
        finally:
            fused.gru_sequence.use_persistent = True
    for a, b, name in zip(res[0], res[1], ["out"] + list(arrs)):
        assert np.allclose(a, b, rtol=2e-5, atol=2e-5 * np.abs(b).max()), name


def test_integration_snippet_runs_without_torch():
    """The ctypes-only binding of INTEGRATION.md drives the library in a process that loads neither PyTorch
    nor this package (device runtime, allocator, copies and GEMM all come from libpdnhip.so)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "integration_snippet.py")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "torch loaded: False" in r.stdout


# ---------------------------------------------------------------------------------------
# row-resident projection GEMM (csrc/gemm_rowres.hip): A rows in registers, B streamed through LDS
@pytest.mark.parametrize("M,N,trans,extras", [(8192, 768, 0, 0), (8200, 864, 0, 1), (8192 + 40, 160, 1, 2),
                                              (8192, 768, 1, 0), (300, 96, 0, 3), (16384, 1120, 0, 0)])
def test_gemm_row_resident_entry_point(hip, M, N, trans, extras):
    # extras: 1 = bias, 2 = residual, 3 = both; N = 160 / 1120 leave a partial last 96-column chunk, the
    # odd M values a partial last 32-row block
    from pydynet_amd import _lib
    L = _lib.lib()
    K = 288
    rng = np.random.default_rng(M + N + trans)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = 0.1 * rng.standard_normal((N, K) if trans else (K, N), dtype=np.float32)
    bias = rng.standard_normal(N, dtype=np.float32) if extras & 1 else None
    res = rng.standard_normal((M, N), dtype=np.float32) if extras & 2 else None
    X, W, Y = hip.from_numpy(x), hip.from_numpy(w), hip.empty((M, N), np.float32)
    Bv = hip.from_numpy(bias) if bias is not None else None
    Rv = hip.from_numpy(res) if res is not None else None
    assert L.query("pdn_gemm_rowres_supported", M, N, K, K, w.shape[1], N, trans)
    L.call("pdn_gemm_rowres_f32", X._ptr, W._ptr, Y._ptr, Bv._ptr if Bv is not None else None,
           Rv._ptr if Rv is not None else None, M, N, K, K, w.shape[1], N, trans, hip.stream())
    ref = x.astype(np.float64) @ (w.T if trans else w).astype(np.float64)
    if bias is not None: ref = ref + bias
    if res is not None: ref = ref + res
    assert rel_err(Y.get(), ref) < 1e-5
    assert not L.query("pdn_gemm_rowres_supported", M, N, 256, 256, w.shape[1], N, trans)     # K = 288 only


def test_gemm_row_resident_dispatch_matches_tiled_kernel(hip):
    # pdn_gemm_f32 routes tall K = 288 products with wide outputs to the row-resident kernel; its k-order per
    # output element is the tiled kernel's, so the two agree bit for bit -- also for the batch that is
    # really one product (x against equally spaced weights, results side by side: fused QKV / gate | up)
    K, M = 288, 8192
    rng = np.random.default_rng(5)
    x = rng.standard_normal((M, K), dtype=np.float32)
    X = hip.from_numpy(x)
    for trans, N in ((0, 768), (1, 768)):
        w = 0.1 * rng.standard_normal((N, K) if trans else (K, N), dtype=np.float32)
        W = hip.from_numpy(w)
        Wv = W.T if trans else W
        y1 = hip.matmul(X, Wv).get()
        os.environ["PDN_GEMM_NO_ROWRES"] = "1"
        try:
            y0 = hip.matmul(X, Wv).get()
        finally:
            del os.environ["PDN_GEMM_NO_ROWRES"]
        assert np.array_equal(y0, y1)
        assert rel_err(y1, x.astype(np.float64) @ (w.T if trans else w).astype(np.float64)) < 1e-5
    # three (288 x 288) weights, descending addresses, one packed (M, 864) result
    ws = 0.1 * rng.standard_normal((3, K, K), dtype=np.float32)
    flat = hip.from_numpy(ws.reshape(-1).copy())
    views = [flat[(2 - i) * K * K:(3 - i) * K * K].reshape(K, K) for i in range(3)]
    stack = hip.stacked_view(views)
    packed = hip.empty((M, 3 * K), np.float32)
    blocks = hip.ndarray(packed._buf, packed._ptr, (3, M, K), (K, 3 * K, 1), packed.dtype)
    hip.gemm(X, stack, blocks)
    got = packed.get()
    for i in range(3):
        assert rel_err(got[:, i * K:(i + 1) * K], x.astype(np.float64) @ ws[2 - i].astype(np.float64)) < 1e-5


def test_gemm_row_resident_packed_gate_up_blocks(hip):
    # the FFN's gate | up projections: two (288 x 768) weights, equally spaced, ONE launch on the `blocks` entry
    # of the row-resident kernel, results side by side in a packed (M, 1536) buffer (fused.gate_up_swiglu)
    K, M, F = 288, 8192 + 32, 768
    rng = np.random.default_rng(11)
    x = rng.standard_normal((M, K), dtype=np.float32)
    ws = 0.1 * rng.standard_normal((2, K, F), dtype=np.float32)
    X = hip.from_numpy(x)
    buf = hip.from_numpy(ws)
    stack = hip.stacked_view([buf[0], buf[1]])
    assert stack is not None
    packed = hip.empty((M, 2 * F), np.float32)
    blocks = hip.ndarray(packed._buf, packed._ptr, (2, M, F), (F, 2 * F, 1), packed.dtype)
    from pydynet_amd import _lib
    L = _lib.lib()
    import ctypes
    L.call("pdn_gemm_prof_enable", 1)
    hip.gemm(X, stack, blocks)
    ms, fl, n = (ctypes.c_double * 5)(), (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
    L.call("pdn_gemm_prof_enable", 0)
    L.call("pdn_gemm_prof_collect_families", ms, fl, n)
    if type(L).__name__ != "EmulatedLib":                 # (the CPU emulation of the ABI has no kernel families)
        assert n[2] == 1 and sum(n) == 1                  # one launch, on the row-resident kernel
    got = packed.get()
    for i in range(2):
        assert rel_err(got[:, i * F:(i + 1) * F], x.astype(np.float64) @ ws[i].astype(np.float64)) < 1e-5
    # bit-identical to the tiled kernel on the same packing
    os.environ["PDN_GEMM_NO_ROWRES"] = "1"
    try:
        packed0 = hip.empty((M, 2 * F), np.float32)
        blocks0 = hip.ndarray(packed0._buf, packed0._ptr, (2, M, F), (F, 2 * F, 1), packed0.dtype)
        hip.gemm(X, stack, blocks0)
    finally:
        del os.environ["PDN_GEMM_NO_ROWRES"]
    assert np.array_equal(packed0.get(), got)


def test_lm_head_forward_row_resident_with_bias_full_vocab(hip):
    # lm_head forward of the benchmark: (tokens, 288) @ (288, 32000) + bias on the row-resident kernel
    # (32000 = 333 * 96 + 32: the last column chunk is partial), sampled rows / columns against float64
    from pydynet_amd import _lib
    import ctypes
    L = _lib.lib()
    M, K, N = 8192, 288, 32000
    rng = np.random.default_rng(12)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = 0.05 * rng.standard_normal((K, N), dtype=np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    X, W, Bv = hip.from_numpy(x), hip.from_numpy(w), hip.from_numpy(b)
    Y = hip.empty((M, N), np.float32)
    L.call("pdn_gemm_prof_enable", 1)
    hip.gemm(X, W, Y, bias=Bv)
    ms, fl, n = (ctypes.c_double * 5)(), (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
    L.call("pdn_gemm_prof_enable", 0)
    L.call("pdn_gemm_prof_collect_families", ms, fl, n)
    if type(L).__name__ != "EmulatedLib":
        assert n[2] == 1 and sum(n) == 1
    y = Y.get()
    rows = np.concatenate([rng.integers(0, M, 24), [0, M - 1]])
    ref = x[rows].astype(np.float64) @ w.astype(np.float64) + b
    assert rel_err(y[rows], ref) < 1e-5
    cols = np.concatenate([rng.integers(0, N, 24), [0, 31967, 31968, N - 1]])      # both sides of the partial chunk
    refc = x.astype(np.float64) @ w[:, cols].astype(np.float64) + b[cols]
    assert rel_err(y[:, cols], refc) < 1e-5
    os.environ["PDN_GEMM_NO_ROWRES"] = "1"
    try:
        Y0 = hip.empty((M, N), np.float32)
        hip.gemm(X, W, Y0, bias=Bv)
    finally:
        del os.environ["PDN_GEMM_NO_ROWRES"]
    assert np.array_equal(Y0.get(), y)


# ---------------------------------------------------------------------------------------
# output-resident GEMM (csrc/gemm_outres.hip): 32 x 288 outputs per wave in accumulators, A straight into
# MFMA operand registers, B through LDS
@pytest.mark.parametrize("M,K,trans,extras", [(512, 768, 0, 0), (300, 864, 1, 2), (8192 + 40, 1536, 1, 3),
                                              (256, 32, 0, 1), (1024, 3200, 1, 0), (16384 + 40, 864, 1, 3)])
def test_gemm_output_resident_entry_point(hip, M, K, trans, extras):
    # extras: 1 = bias, 2 = residual, 3 = both; odd M values leave partial 32-row blocks and partial workgroups
    from pydynet_amd import _lib
    L = _lib.lib()
    N = 288
    rng = np.random.default_rng(M + K + trans)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = 0.1 * rng.standard_normal((N, K) if trans else (K, N), dtype=np.float32)
    bias = rng.standard_normal(N, dtype=np.float32) if extras & 1 else None
    res = rng.standard_normal((M, N), dtype=np.float32) if extras & 2 else None
    X, W, Y = hip.from_numpy(x), hip.from_numpy(w), hip.empty((M, N), np.float32)
    Bv = hip.from_numpy(bias) if bias is not None else None
    Rv = hip.from_numpy(res) if res is not None else None
    assert L.query("pdn_gemm_outres_supported", M, N, K, K, w.shape[1], N, trans)
    L.call("pdn_gemm_outres_f32", X._ptr, W._ptr, Y._ptr, Bv._ptr if Bv is not None else None,
           Rv._ptr if Rv is not None else None, M, N, K, K, w.shape[1], N, trans, hip.stream())
    ref = x.astype(np.float64) @ (w.T if trans else w).astype(np.float64)
    if bias is not None: ref = ref + bias
    if res is not None: ref = ref + res
    assert rel_err(Y.get(), ref) < 1e-5
    assert not L.query("pdn_gemm_outres_supported", M, 256, K, K, w.shape[1], 256, trans)     # N = 288 only
    assert not L.query("pdn_gemm_outres_supported", M, N, K + 8, K + 8, w.shape[1] + 8, N, trans)


@pytest.mark.parametrize("M,K,trans,extras", [(16384, 3200, 1, 3), (16384, 1536, 0, 2), (8192 + 40, 4096, 1, 1),
                                              (32768, 1536, 1, 2)])
def test_gemm_output_resident_split_k(hip, M, K, trans, extras):
    # fewer than ~224 row workgroups: K is cut into ranges over grid.y, one (M x 288) slab per range in the workspace,
    # added up (+ bias + residual) in a fixed order.  Against float64 on sampled rows and against the unsplit launch.
    from pydynet_amd import _lib
    import ctypes
    L = _lib.lib()
    N = 288
    rng = np.random.default_rng(M + K + trans)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = 0.1 * rng.standard_normal((N, K) if trans else (K, N), dtype=np.float32)
    bias = rng.standard_normal(N, dtype=np.float32) if extras & 1 else None
    res = rng.standard_normal((M, N), dtype=np.float32) if extras & 2 else None
    X, W = hip.from_numpy(x), hip.from_numpy(w)
    Bv = hip.from_numpy(bias) if bias is not None else None
    Rv = hip.from_numpy(res) if res is not None else None
    nw, kps = ctypes.c_int(), ctypes.c_int()
    splits = L.query("pdn_gemm_outres_plan", M, K, ctypes.byref(nw), ctypes.byref(kps))
    if type(L).__name__ != "EmulatedLib":
        # (8-wave workgroups over ranges where ~256 of them come out -- 16384 x 3200: 4 ranges, 32768 x 1536: 2 -- else 4-wave)
        assert splits >= 2 and kps.value >= 24 and nw.value == (8 if (M, K) in ((16384, 3200), (32768, 1536)) else 4)
    wsb = L.query("pdn_gemm_outres_workspace_bytes", M, K)
    ws, got_b = hip.workspace(max(wsb, 4))
    Y, Y1 = hip.empty((M, N), np.float32), hip.empty((M, N), np.float32)
    args = (Bv._ptr if Bv is not None else None, Rv._ptr if Rv is not None else None, M, N, K, K, w.shape[1], N, trans)
    L.call("pdn_gemm_outres_ws_f32", X._ptr, W._ptr, Y._ptr, *args, ws, got_b, hip.stream())
    L.call("pdn_gemm_outres_f32", X._ptr, W._ptr, Y1._ptr, *args, hip.stream())
    y, y1 = Y.get(), Y1.get()
    assert rel_err(y, y1.astype(np.float64)) < 1e-5                        # same products, another summation order
    rows = np.concatenate([rng.integers(0, M, 48), [0, M - 1]])
    ref = x[rows].astype(np.float64) @ (w.T if trans else w).astype(np.float64)
    if bias is not None: ref = ref + bias
    if res is not None: ref = ref + res[rows]
    assert rel_err(y[rows], ref) < 1e-5
    # the same through pdn_gemm_f32's dispatch (hipnp.gemm hands it a workspace)
    Y2 = hip.empty((M, N), np.float32)
    hip.gemm(X, W.T if trans else W, Y2, bias=Bv, residual=Rv)
    assert rel_err(Y2.get()[rows], ref) < 1e-5


def test_gemm_output_resident_dispatch_matches_tiled_kernel(hip):
    # pdn_gemm_f32 sends tall products that end in 288 columns with K >= 768 to the output-resident kernel (also with
    # the residual of a transformer block folded in); same k-order per element as the tiled kernel: bit-identical
    M, N = 57344, 288
    rng = np.random.default_rng(9)
    for trans, K, with_res in ((0, 768, True), (1, 864, False)):
        x = rng.standard_normal((M, K), dtype=np.float32)
        w = 0.1 * rng.standard_normal((N, K) if trans else (K, N), dtype=np.float32)
        r = rng.standard_normal((M, N), dtype=np.float32) if with_res else None
        X, W = hip.from_numpy(x), hip.from_numpy(w)
        R = hip.from_numpy(r) if with_res else None
        Wv = W.T if trans else W
        y1 = hip.empty((M, N), np.float32)
        hip.gemm(X, Wv, y1, residual=R)
        os.environ["PDN_GEMM_NO_OUTRES"] = "1"
        try:
            y0 = hip.empty((M, N), np.float32)
            hip.gemm(X, Wv, y0, residual=R)
        finally:
            del os.environ["PDN_GEMM_NO_OUTRES"]
        a, b = y0.get(), y1.get()
        assert np.array_equal(a, b)
        ref = x[:512].astype(np.float64) @ (w.T if trans else w).astype(np.float64) + (r[:512] if with_res else 0.0)
        assert rel_err(b[:512], ref) < 1e-5


@pytest.mark.parametrize("N,K,beta", [(8192, 4096, 0.0), (8192 + 32, 8192 + 64, 1.0), (16384, 4096, 1.0)])
def test_gemm_weight_gradient_output_resident(hip, N, K, beta):
    # x^T @ g with 288 output rows and a wide g (the lm_head weight gradient): output-resident TN kernel, K split
    # over the grid, slabs combined with beta * C by the split-K reduce
    rng = np.random.default_rng(N + K)
    x = rng.standard_normal((K, 288), dtype=np.float32)
    g = rng.standard_normal((K, N), dtype=np.float32)
    c0 = rng.standard_normal((288, N), dtype=np.float32)
    X, G, C = hip.from_numpy(x), hip.from_numpy(g), hip.from_numpy(c0.copy())
    hip.gemm(X.T, G, C, beta=beta)
    ref = x.T.astype(np.float64) @ g.astype(np.float64) + beta * c0
    assert rel_err(C.get(), ref) < 2e-5


@pytest.mark.parametrize("nblk,Nb", [(3, 288), (2, 768)])
def test_gemm_packed_weight_gradients_output_resident(hip, nblk, Nb):
    # x^T against the column blocks of one packed (K, nblk * Nb) gradient buffer, one (288, Nb) weight gradient per
    # block accumulated in place (beta = 1): the batch that the fused QKV / gate|up nodes issue; output-resident TN
    # kernel with K split over the grid, slabs in the batched layout of the split-K reduce
    K = 16384
    rng = np.random.default_rng(nblk)
    x = rng.standard_normal((K, 288), dtype=np.float32)
    g = rng.standard_normal((K, nblk * Nb), dtype=np.float32)
    w0 = rng.standard_normal((nblk, 288, Nb), dtype=np.float32)
    X, G = hip.from_numpy(x), hip.from_numpy(g)
    flat = hip.from_numpy(w0.reshape(-1).copy())
    views = [flat[(nblk - 1 - i) * 288 * Nb:(nblk - i) * 288 * Nb].reshape(288, Nb) for i in range(nblk)]   # descending
    stack = hip.stacked_view(views)
    blocks = hip.ndarray(G._buf, G._ptr, (nblk, K, Nb), (Nb, nblk * Nb, 1), G.dtype)
    hip.gemm(X.T, blocks, stack, beta=1.0)
    for i in range(nblk):
        ref = w0[nblk - 1 - i] + x.T.astype(np.float64) @ g[:, i * Nb:(i + 1) * Nb].astype(np.float64)
        assert rel_err(views[i].get(), ref) < 2e-5


def test_gemm_resident_kernels_strided_operands(hip):
    # A as a column block of a wider packed buffer (row stride 864 > K), C as a column block of a wider output:
    # the row-resident and output-resident kernels take leading dimensions, not just contiguous matrices
    M = 57344
    rng = np.random.default_rng(21)
    packed = rng.standard_normal((M, 864), dtype=np.float32)
    P = hip.from_numpy(packed)
    a_view = P[:, 288:576]                                     # (M, 288), row stride 864
    w1 = 0.1 * rng.standard_normal((288, 768), dtype=np.float32)
    out = hip.empty((M, 1536), np.float32)
    c_view = out[:, 768:]                                      # (M, 768), row stride 1536
    hip.gemm(a_view, hip.from_numpy(w1), c_view)               # row-resident (K = 288, N = 768)
    ref = packed[:256, 288:576].astype(np.float64) @ w1.astype(np.float64)
    assert rel_err(out.get()[:256, 768:], ref) < 1e-5
    a2 = P[:, :768]                                            # (M, 768), row stride 864
    w2 = 0.1 * rng.standard_normal((768, 288), dtype=np.float32)
    out2 = hip.empty((M, 576), np.float32)
    hip.gemm(a2, hip.from_numpy(w2), out2[:, 288:])            # output-resident (N = 288, K = 768)
    ref2 = packed[-256:, :768].astype(np.float64) @ w2.astype(np.float64)
    assert rel_err(out2.get()[-256:, 288:], ref2) < 1e-5
