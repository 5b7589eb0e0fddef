"""The tile-piece row-resident kernel (csrc/gemm_rowtile.hip, round 5) against the chunk kernel it replaces
(csrc/gemm_rowres.hip) and against float64: same k-order per output element, so every entry point must give
BIT-identical results on either kernel -- plain projections (`x @ W`, `grad @ W^T`: pydynet/core/tensor.py:657-676), gate | up
+ SwiGLU and its backward (llm/llama/model.py:56-58), q | k | v + RoPE (model.py:23-44), the vocabulary projection with row
maxima (nn/functional.py:364-381).  Shapes include ragged row blocks (GUARD instantiations), odd tile counts, column
ranges split over grid.y and weights in either memory order.  `pdn_gemm_rowtile_mode(2)` forces the new kernel onto
shapes the default selection leaves to the chunk kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 288


def _lib_hp():
    from pydynet_amd import _lib, hipnp
    return _lib.lib(), hipnp


def _both(run):
    """run() -> list of device arrays; executed on the chunk kernel (mode 0) and the tile-piece kernel (mode 2)."""
    L, hp = _lib_hp()
    prev = L.query("pdn_gemm_rowtile_mode", 0)
    try:
        old = [a.get().copy() for a in run()]
        L.query("pdn_gemm_rowtile_mode", 2)
        new = [a.get().copy() for a in run()]
    finally:
        L.query("pdn_gemm_rowtile_mode", prev)
    return old, new


def _same(old, new, names):
    for o, n, name in zip(old, new, names):
        assert o.shape == n.shape
        bad = np.flatnonzero(o.view(np.uint32).ravel() != n.view(np.uint32).ravel())
        assert bad.size == 0, (name, bad.size, bad[:8], o.ravel()[bad[:4]], n.ravel()[bad[:4]])


def _close(a, b, what, rt=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err, scale = float(np.abs(a - b).max()), max(float(np.abs(b).max()), 1e-30)
    assert err <= 1e-7 + rt * scale, (what, err, scale)


def _stack(hp, mats):
    buf = hp.empty((len(mats),) + mats[0].shape, np.float32)
    views = []
    for i, m in enumerate(mats):
        buf[i] = hp.from_numpy(m)
        views.append(buf[i])
    return buf, views, int(np.prod(mats[0].shape))


@pytest.mark.parametrize("M,N,bt,bias", [(256, 96, 0, True), (300, 864, 0, False), (1000, 160, 0, True), (300, 768, 1, False),
                                         (256, 1056, 1, True), (2048, 2080, 0, True)])
def test_plain_projection_bit_identical(hip, M, N, bt, bias):
    L, hp = _lib_hp()
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (0.1 * rng.standard_normal((N, K) if bt else (K, N))).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    xd, wd, bd = hp.from_numpy(x), hp.from_numpy(w), hp.from_numpy(b)
    assert L.query("pdn_gemm_rowres_supported", M, N, K, K, K if bt else N, N, bt)

    def run():
        c = hp.empty((M, N), np.float32)
        c[...] = 7.0
        L.call("pdn_gemm_rowres_f32", xd._ptr, wd._ptr, c._ptr, bd._ptr if bias else None, None, M, N, K, K, K if bt else N, N,
               bt, hp.stream())
        return [c]

    old, new = _both(run)
    _same(old, new, ["C"])
    ref = x.astype(np.float64) @ (w.T if bt else w).astype(np.float64) + (b if bias else 0.0)
    _close(new[0], ref, "x @ W")


@pytest.mark.parametrize("M,N,bt,bias", [(256, 288, 0, False), (300, 288, 1, True), (2048, 864, 0, True), (1000, 96, 1, False)])
def test_projection_with_residual_bit_identical(hip, M, N, bt, bias):
    """C = x @ W (+ bias) + residual: the residual rows ride in the drain of the tile-piece kernel (EPI 6), eight row
    steps ahead of their add -- the transformer's `x + sublayer(x)` (llm/llama/model.py:140-150) and the engine's
    "add to the gradient this input already holds" (tensor.py:371)."""
    L, hp = _lib_hp()
    rng = np.random.default_rng(M + N + 1)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (0.1 * rng.standard_normal((N, K) if bt else (K, N))).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    xd, wd, bd, rd = hp.from_numpy(x), hp.from_numpy(w), hp.from_numpy(b), hp.from_numpy(res)

    def run():
        c = hp.empty((M, N), np.float32)
        c[...] = 7.0
        L.call("pdn_gemm_rowres_f32", xd._ptr, wd._ptr, c._ptr, bd._ptr if bias else None, rd._ptr, M, N, K, K, K if bt else N, N,
               bt, hp.stream())
        return [c]

    old, new = _both(run)
    _same(old, new, ["C"])
    ref = x.astype(np.float64) @ (w.T if bt else w).astype(np.float64) + (b if bias else 0.0) + res
    _close(new[0], ref, "x @ W + residual")


def test_attention_output_projection_takes_the_tile_piece_kernel(hip):
    """The 288 x 288 products of a block (ctx Wo + x, dz Wo^T: llm/llama/model.py:121, tensor.py:670) through pdn_gemm_f32 at
    the benchmark row count: routed to the tile-piece kernel (counter), equal to the tiled kernel's result."""
    import ctypes
    L, hp = _lib_hp()
    M = 16384
    rng = np.random.default_rng(3)
    x = hp.from_numpy(rng.standard_normal((M, K), dtype=np.float32))
    w = hp.from_numpy((0.1 * rng.standard_normal((K, K))).astype(np.float32))
    res = hp.from_numpy(rng.standard_normal((M, K), dtype=np.float32))
    outs = []
    for mode in (0, 1):
        prev = L.query("pdn_gemm_rowtile_mode", mode)
        try:
            L.call("pdn_kernel_counters", None, 0, 1)
            y, dx = hp.empty((M, K)), hp.empty((M, K))
            hp.gemm(x, w, y, residual=res)
            hp.gemm(x, w.T, dx)
            buf = (ctypes.c_int64 * 16)()
            L.call("pdn_kernel_counters", buf, 16, 1)
            outs.append((y.get(), dx.get(), int(buf[1])))
        finally:
            L.query("pdn_gemm_rowtile_mode", prev)
    assert outs[0][2] == 0 and outs[1][2] == 2, (outs[0][2], outs[1][2])
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_blocks_projection_bit_identical(hip):
    """x [W0 | W1 | W2] with the three matrices equally spaced (the packed q | k | v projection), ragged rows."""
    L, hp = _lib_hp()
    M, D = 8200, 288                 # (pdn_gemm_f32 routes batched projections to the row-resident kernels from 8192 rows)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((M, K)).astype(np.float32)
    ws = [(0.1 * rng.standard_normal((K, D))).astype(np.float32) for _ in range(3)]
    xd = hp.from_numpy(x)
    buf, views, stride = _stack(hp, ws)

    def run():
        out = hp.empty((M, 3 * D), np.float32)
        blocks = hp.ndarray(out._buf, out._ptr, (3, M, D), (D, 3 * D, 1), out.dtype)
        hp.gemm(xd, buf, blocks)
        return [out]

    old, new = _both(run)
    _same(old, new, ["qkv"])
    _close(new[0], np.concatenate([x.astype(np.float64) @ w for w in ws], 1), "x [W0|W1|W2]")


@pytest.mark.parametrize("M,F,up_first", [(512, 192, False), (300, 768, False), (256, 96, True), (2304, 288, False)])
def test_gateup_swiglu_both_directions_bit_identical(hip, M, F, up_first):
    L, hp = _lib_hp()
    rng = np.random.default_rng(M + F)
    x = rng.standard_normal((M, K)).astype(np.float32)
    wg = (0.08 * rng.standard_normal((K, F))).astype(np.float32)
    wu = (0.08 * rng.standard_normal((K, F))).astype(np.float32)
    wdn = (0.08 * rng.standard_normal((F, K))).astype(np.float32)
    dy = rng.standard_normal((M, K)).astype(np.float32)
    if up_first:
        buf, (du, dg), stride = _stack(hp, [wu, wg])
        stride = -stride
    else:
        buf, (dg, du), stride = _stack(hp, [wg, wu])
    xd, dyd, wdd = hp.from_numpy(x), hp.from_numpy(dy), hp.from_numpy(wdn)

    def run():
        gu, h, dgu = hp.empty((M, 2 * F), np.float32), hp.empty((M, F), np.float32), hp.empty((M, 2 * F), np.float32)
        for a in (gu, h, dgu):
            a[...] = -3.0
        L.call("pdn_gateup_swiglu_fwd_f32", xd._ptr, dg._ptr, stride, gu._ptr, h._ptr, M, F, K, K, hp.stream())
        L.call("pdn_swiglu_bwd_gemm_f32", dyd._ptr, wdd._ptr, gu._ptr, dgu._ptr, M, F, K, K, hp.stream())
        return [gu, h, dgu]

    old, new = _both(run)
    _same(old, new, ["gate|up", "h", "d gate|up"])
    g64, u64 = x.astype(np.float64) @ wg, x.astype(np.float64) @ wu
    _close(new[0][:, :F], g64, "gate")
    _close(new[0][:, F:], u64, "up")
    _close(new[1], g64 / (1 + np.exp(-g64)) * u64, "h")
    gs, us = new[0][:, :F].astype(np.float64), new[0][:, F:].astype(np.float64)
    s = 1 / (1 + np.exp(-gs))
    dh = dy.astype(np.float64) @ wdn.astype(np.float64).T
    _close(new[2][:, :F], dh * us * s * (1 + gs * (1 - s)), "d gate")
    _close(new[2][:, F:], dh * gs * s, "d up")


def _tables(Lq, hd):
    inv = 1.0 / (10000 ** (np.arange(0, hd, 2)[: hd // 2] / hd))
    fr = np.outer(np.arange(Lq), inv)
    return np.cos(fr).astype(np.float32), np.sin(fr).astype(np.float32)


def _rope_ref(y, cos, sin, Lq, hd):
    M, D = y.shape
    pos = np.arange(M) % Lq
    yh = y.reshape(M, D // hd, hd // 2, 2)
    c, s = cos[pos][:, None, :], sin[pos][:, None, :]
    out = np.empty_like(yh)
    out[..., 0] = yh[..., 0] * c - yh[..., 1] * s
    out[..., 1] = yh[..., 0] * s + yh[..., 1] * c
    return out.reshape(M, D)


@pytest.mark.parametrize("B,Lq,hd", [(4, 64, 48), (3, 32, 96), (9, 256, 48), (5, 96, 32)])
def test_qkv_rope_bit_identical(hip, B, Lq, hd):
    L, hp = _lib_hp()
    D, M = 288, B * Lq
    rng = np.random.default_rng(B * Lq + hd)
    x = rng.standard_normal((M, K)).astype(np.float32)
    ws = [(0.08 * rng.standard_normal((K, D))).astype(np.float32) for _ in range(3)]
    cos, sin = _tables(Lq, hd)
    assert L.query("pdn_qkv_rope_supported", M, D, K, Lq, hd)
    buf, views, stride = _stack(hp, ws)
    tab = hp.empty((Lq, hd, 2), np.float32)
    cd, sd, xd = hp.from_numpy(cos), hp.from_numpy(sin), hp.from_numpy(x)
    L.call("pdn_rope_table_f32", cd._ptr, sd._ptr, tab._ptr, Lq, hd, hp.stream())

    def run():
        qkv = hp.empty((M, 3 * D), np.float32)
        qkv[...] = 11.0
        L.call("pdn_qkv_rope_fwd_f32", xd._ptr, views[0]._ptr, stride, qkv._ptr, tab._ptr, M, D, K, Lq, hd, K, hp.stream())
        return [qkv]

    old, new = _both(run)
    _same(old, new, ["q|k|v"])
    x64, c64, s64 = x.astype(np.float64), cos.astype(np.float64), sin.astype(np.float64)
    _close(new[0][:, :D], _rope_ref(x64 @ ws[0], c64, s64, Lq, hd), "rotated q")
    _close(new[0][:, D:2 * D], _rope_ref(x64 @ ws[1], c64, s64, Lq, hd), "rotated k")
    _close(new[0][:, 2 * D:], x64 @ ws[2], "v")


@pytest.mark.parametrize("M,V,bias", [(300, 1056, True), (256, 4000, False), (2048, 992, True), (1024, 32000, True)])
def test_rowmax_projection_bit_identical(hip, M, V, bias):
    """logits + the row maxima (the maxima come as `parts` vectors: one per column range of the launch)."""
    L, hp = _lib_hp()
    rng = np.random.default_rng(M + V)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (0.1 * rng.standard_normal((K, V))).astype(np.float32)
    b = rng.standard_normal(V).astype(np.float32)
    xd, wd, bd = hp.from_numpy(x), hp.from_numpy(w), hp.from_numpy(b)
    assert L.query("pdn_linear_rowmax_supported", M, V, K)

    def run():
        parts = L.query("pdn_linear_rowmax_parts", M, V, K)
        assert parts >= 1
        logits, mx = hp.empty((M, V), np.float32), hp.empty((parts, M), np.float32)
        logits[...] = 5.0
        mx[...] = 99.0
        L.call("pdn_linear_rowmax_fwd_f32", xd._ptr, wd._ptr, bd._ptr if bias else None, logits._ptr, mx._ptr, M, V, K, K, V, V,
               hp.stream())
        m = hp.from_numpy(mx.get().max(0))
        return [logits, m]

    old, new = _both(run)
    _same(old, new, ["logits", "row maxima"])
    assert np.array_equal(new[1], new[0].max(1))
    _close(new[0], x.astype(np.float64) @ w.astype(np.float64) + (b if bias else 0.0), "logits")


def test_swiglu_backward_in_place_stays_correct(hip):
    """d[gate | up] written OVER the saved [gate | up] (an in-place caller): the entry point must not take the tile-piece
    kernel, whose first (dummy) drain writes a tile before its gate / up rows were read; results equal the out-of-place call."""
    L, hp = _lib_hp()
    M, F = 2560, 768
    rng = np.random.default_rng(77)
    dy = hp.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    wdn = hp.from_numpy((0.08 * rng.standard_normal((F, K))).astype(np.float32))
    gu_h = rng.standard_normal((M, 2 * F)).astype(np.float32)
    gu, out = hp.from_numpy(gu_h), hp.empty((M, 2 * F), np.float32)
    L.call("pdn_swiglu_bwd_gemm_f32", dy._ptr, wdn._ptr, gu._ptr, out._ptr, M, F, K, K, hp.stream())
    inplace = hp.from_numpy(gu_h)
    L.call("pdn_swiglu_bwd_gemm_f32", dy._ptr, wdn._ptr, inplace._ptr, inplace._ptr, M, F, K, K, hp.stream())
    assert np.array_equal(out.get(), inplace.get())
