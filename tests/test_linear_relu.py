"""`relu(linear(x))` as one product with one gradient bit per element (csrc/gemm.hip: pdn_linear_relu_fwd_f32,
pdn_linear_dx_masked_f32, pdn_relu_mask_bwd_f32; core/fused/dense.py: linear_relu) -- the Linear -> ReLU -> Linear -> ReLU ->
Linear chain of examples/pydynet/mnist.py:70-78 (nn/functional.py:31-32 relu = maximum(0., x); tensor.py:808-814: the
gradient passes where out == x, i.e. pre-activation >= 0, INCLUDING exactly 0).

Kernel level against float64 NumPy statements of those lines; node level against the same modules with the fusion
switched off (`fused.linear.defer = False`: Linear and ReLU as the two nodes the reference builds).  Also the two small
kernels the MLP step needed: cross entropy over a handful of classes (one thread per row) and the wide column sum.
Tolerance 1e-4 relative to the tensor's largest entry (north_star).  Runs on the emulated C ABI and (-m gpu) on MI355X.
"""
import ctypes

import numpy as np
import pytest

import pydynet_amd as pdn
import pydynet_amd.nn as nn
import pydynet_amd.nn.functional as F
from pydynet_amd.core import fused
from tests.conftest import device_variants

RT = 1e-4


def host(x):
    return x.numpy() if isinstance(x, pdn.Tensor) else (x if isinstance(x, np.ndarray) else x.get())


def close(a, b, what, rt=RT):
    a, b = np.asarray(host(a), np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a - b).max())
    assert err <= 1e-7 + rt * scale, (what, err, scale)


def _lib_hp():
    from pydynet_amd import _lib, hipnp
    return _lib.lib(), hipnp


def _counters(L):
    buf = (ctypes.c_int64 * 19)()
    L.call("pdn_kernel_counters", buf, 19, 1)
    return list(buf)


def _bits(words, rows, cols):
    """(rows x cols) booleans from the library's words: bit c of word [row][col // 32] is column 32 * (col // 32) + c."""
    w = np.ascontiguousarray(host(words)).view(np.uint8).reshape(rows, cols // 8)
    return np.unpackbits(w, axis=1, bitorder="little").astype(bool)


# ---- kernel level ------------------------------------------------------------------------------------------------
def _fwd_case(M, N, K, seed, bias=True, w_transposed=False):
    L, hp = _lib_hp()
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) * 0.1 if bias else None
    x[3] = 0.0                                   # a blank input row: pre-activation == bias exactly ...
    if b is not None:
        b[5] = 0.0                               # ... which is exactly 0 in column 5: the gradient must PASS there
    xd = hp.from_numpy(x)
    wd = hp.from_numpy(np.ascontiguousarray(w.T)).T if w_transposed else hp.from_numpy(w)
    bd = hp.from_numpy(b) if b is not None else None
    h = hp.empty((M, N), np.float32)
    bits = hp.empty((M, N // 32), np.float32)
    L.call("pdn_linear_relu_fwd_f32", xd._ptr, K, wd._ptr, wd._strides[0], wd._strides[1], bd._ptr if bd is not None else None,
           h._ptr, N, bits._ptr, M, N, K, hp.stream())
    # the same product without the relu, by the library itself: the fused store must be max(0, .) of exactly these values
    z = hp.empty((M, N), np.float32)
    hp.gemm(xd, wd, z, bias=bd)
    zh, hh = host(z), host(h)
    z64 = x.astype(np.float64) @ w.astype(np.float64) + (b.astype(np.float64) if b is not None else 0.0)
    close(hh, np.maximum(0.0, z64), "relu(x W + b)")
    got = _bits(bits, M, N)
    # bits: wherever the float32 pre-activation is clearly off zero they must equal its sign; on the blank row exactly
    sure = np.abs(z64) > 1e-5 * np.abs(z64).max()
    assert np.array_equal(got[sure], (z64 >= 0)[sure]), "gradient bits"
    assert np.array_equal(hh > 0, got & (hh > 0)) and not np.any(hh[~got] != 0), "bits vs stored activations"
    if b is not None:
        assert got[3, 5] and hh[3, 5] == 0.0, "a pre-activation of exactly 0 passes the gradient (maximum's out == x)"
        assert np.array_equal(got[3], b >= 0), "blank row: bits of the bias"
    return (L, hp), (xd, wd, bd, h, bits), (x, w, b, z64, got)


def check_linear_relu_forward_tiles(device):
    for (M, N, K, seed, bias, wt) in ((256, 128, 64, 0, True, False), (1000, 96, 50, 1, True, False),
                                      (130, 32, 784, 2, False, False), (512, 1024, 256, 3, True, True),
                                      (77, 160, 33, 4, True, False)):
        _fwd_case(M, N, K, seed, bias, wt)


device_variants(globals(), check_linear_relu_forward_tiles)


def check_linear_dx_masked_and_mask_bwd(device):
    for (M, fin, fout, seed, existing) in ((256, 128, 64, 0, False), (1000, 96, 10, 1, True), (300, 1024, 10, 2, False),
                                           (129, 32, 200, 3, True), (4096, 1024, 1024, 4, False), (8200, 256, 512, 5, True)):
        L, hp = _lib_hp()
        rng = np.random.default_rng(seed)
        g = rng.standard_normal((M, fout)).astype(np.float32)
        w = rng.standard_normal((fin, fout)).astype(np.float32)
        ex = rng.standard_normal((M, fin)).astype(np.float32) if existing else None
        keep = rng.random((M, fin)) < 0.5
        words = np.packbits(keep, axis=1, bitorder="little").view(np.float32).reshape(M, fin // 32)
        gd, wd, md = hp.from_numpy(g), hp.from_numpy(w), hp.from_numpy(np.ascontiguousarray(words))
        exd = hp.from_numpy(ex) if ex is not None else None
        dx = hp.empty((M, fin), np.float32)
        nb = (M + 31) // 32
        parts = hp.empty((nb, fin), np.float32)
        _counters(L)
        L.call("pdn_linear_dx_masked_f32", gd._ptr, fout, wd._ptr, wd._strides[0], wd._strides[1], dx._ptr, fin,
               exd._ptr if exd is not None else None, md._ptr, parts._ptr, M, fin, fout, hp.stream())
        assert _counters(L)[17] == 1
        want = g.astype(np.float64) @ w.astype(np.float64).T + (ex.astype(np.float64) if ex is not None else 0.0)
        want = np.where(keep, want, 0.0)
        close(dx, want, "mask o (g W^T + existing)")
        assert not np.any(host(dx)[~keep] != 0)
        pad = np.zeros((nb * 32, fin))
        pad[:M] = want
        close(parts, pad.reshape(nb, 32, fin).sum(1), "column sums per 32-row band", rt=2e-5)
        dx2 = hp.empty((M, fin), np.float32)             # without partials: the same stores
        L.call("pdn_linear_dx_masked_f32", gd._ptr, fout, wd._ptr, wd._strides[0], wd._strides[1], dx2._ptr, fin,
               exd._ptr if exd is not None else None, md._ptr, None, M, fin, fout, hp.stream())
        assert np.array_equal(host(dx2), host(dx))
        dz = hp.empty((M, fin), np.float32)
        src = hp.from_numpy(want.astype(np.float32))
        L.call("pdn_relu_mask_bwd_f32", src._ptr, md._ptr, dz._ptr, M, fin, hp.stream())
        assert np.array_equal(host(dz), np.where(keep, want.astype(np.float32), np.float32(0)))


device_variants(globals(), check_linear_dx_masked_and_mask_bwd)


# ---- node level --------------------------------------------------------------------------------------------------
class _MLP(nn.Module):
    def __init__(self, sizes):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(a, b, dtype=np.float32) for a, b in zip(sizes[:-1], sizes[1:])])

    def forward(self, x):
        for lin in self.layers[:-1]:
            x = F.relu(lin(x))
        return self.layers[-1](x)


def _mlp_step(device, sizes, batch, fuse, seed=0, twice_consumed=False):
    fused.linear.defer = fuse
    saved_rows, fused.linear_relu.min_rows = fused.linear_relu.min_rows, 1      # (small test batches take the fused node too)
    try:
        np.random.seed(seed)
        model = _MLP(sizes).to(device)
        rng = np.random.default_rng(seed + 1)
        x = pdn.Tensor(rng.standard_normal((batch, sizes[0])).astype(np.float32), device=device)
        y = pdn.Tensor(rng.integers(0, sizes[-1], batch), device=device, dtype=np.int64)
        kinds = []
        h = x
        for lin in model.layers[:-1]:
            h = F.relu(lin(h))
            kinds.append(type(h).__name__)
        out = model.layers[-1](h)
        if twice_consumed:
            out = out + 0.01 * h[:, :sizes[-1]]          # a second, non-linear consumer of the last hidden activation
        loss = F.cross_entropy_loss(out, y)
        loss.backward()
        grads = [host(p.grad) for p in model.parameters()]
        return float(loss.item()), grads, kinds
    finally:
        fused.linear.defer = True
        fused.linear_relu.min_rows = saved_rows


def check_mlp_chain_fused_equals_unfused(device):
    L, _ = _lib_hp()
    for sizes, batch in (((784, 1024, 1024, 10), 96), ((40, 64, 32, 7), 50)):
        _counters(L)
        l1, g1, k1 = _mlp_step(device, sizes, batch, True)
        c = _counters(L)
        assert k1 == ["linear_relu", "linear_relu"], k1
        assert c[16] == 2 and c[17] == 2, ("relu epilogue / masked input gradient launches", c[16:18])
        l0, g0, k0 = _mlp_step(device, sizes, batch, False)
        assert k0 == ["relu", "relu"], k0
        assert abs(l1 - l0) <= RT * abs(l0)
        for a, b in zip(g1, g0):
            close(a, b, "parameter gradient, fused vs Linear and ReLU as two nodes")


device_variants(globals(), check_mlp_chain_fused_equals_unfused)


def check_hidden_activation_with_a_second_consumer(device):
    """A gradient that is not (only) a masked input-gradient product gets the bits applied by the node itself."""
    L, _ = _lib_hp()
    l1, g1, _ = _mlp_step(device, (48, 64, 32, 8), 40, True, twice_consumed=True)
    l0, g0, _ = _mlp_step(device, (48, 64, 32, 8), 40, False, twice_consumed=True)
    assert abs(l1 - l0) <= RT * abs(l0)
    for a, b in zip(g1, g0):
        close(a, b, "parameter gradient with a second consumer of h")


device_variants(globals(), check_hidden_activation_with_a_second_consumer)


def check_deferred_linear_is_an_ordinary_node_for_other_consumers(device):
    np.random.seed(0)
    lin = nn.Linear(64, 96, dtype=np.float32).to(device)
    x = pdn.Tensor(np.random.randn(33, 64).astype(np.float32), device=device)
    assert type(lin(x)) is fused.linear and lin(x)._pending is None, "below linear_relu.min_rows the product runs at construction"
    saved_rows, fused.linear_relu.min_rows = fused.linear_relu.min_rows, 1
    try:
        z = lin(x)
    finally:
        fused.linear_relu.min_rows = saved_rows
    assert type(z) is fused.linear and z._pending is not None and z.shape == (33, 96) and z.dtype == np.float32
    y = (z * z).sum()
    assert z._pending is None
    y.backward()
    w, b = host(lin.weight.data), host(lin.bias.data)
    z64 = host(x.data).astype(np.float64) @ w.astype(np.float64) + b.astype(np.float64)
    close(z, z64, "deferred linear, materialised by another consumer")
    close(lin.weight.grad, host(x.data).astype(np.float64).T @ (2 * z64), "its weight gradient")


device_variants(globals(), check_deferred_linear_is_an_ordinary_node_for_other_consumers)


# ---- the two small kernels of the MLP step -----------------------------------------------------------------------
def check_cross_entropy_few_classes(device):
    L, hp = _lib_hp()
    for rows, V in ((4096, 10), (2049, 17), (1024, 32), (1500, 3)):
        rng = np.random.default_rng(rows)
        x = (rng.standard_normal((rows, V)) * 3).astype(np.float32)
        t = rng.integers(0, V, rows)
        t[::7] -= V                                          # negative targets count from the end
        xt = pdn.Tensor(x, device=device, requires_grad=True)
        _counters(L)
        loss = F.cross_entropy_loss(xt, pdn.Tensor(t, device=device, dtype=np.int64))
        loss.backward()
        assert _counters(L)[18] == 1, "one thread per row kernel"
        x64 = x.astype(np.float64)
        lse = np.log(np.exp(x64 - x64.max(1, keepdims=True)).sum(1)) + x64.max(1)
        tt = np.where(t < 0, t + V, t)
        assert abs(float(loss.item()) - (lse - x64[np.arange(rows), tt]).mean()) <= RT * abs(lse.mean())
        want = np.exp(x64 - lse[:, None])
        want[np.arange(rows), tt] -= 1.0
        close(xt.grad, want / rows, "dlogits")


device_variants(globals(), check_cross_entropy_few_classes)


@pytest.mark.gpu
def test_wide_column_sum_gpu(hip):
    """reduce_colsum4_kernel (four columns per thread) against float64, incl. a row count that is not a multiple of the
    unroll and a view with a row stride."""
    hp = hip
    rng = np.random.default_rng(0)
    for rows, cols in ((65536, 1024), (4099, 256), (8192, 260), (5000, 2048)):
        x = rng.standard_normal((rows, cols)).astype(np.float32)
        close(hp.from_numpy(x).sum(0), x.astype(np.float64).sum(0), f"column sums {rows} x {cols}", rt=2e-5)
        close(hp.from_numpy(x).mean(0), x.astype(np.float64).mean(0), "column means", rt=2e-5)
    big = hp.from_numpy(rng.standard_normal((4100, 512)).astype(np.float32))
    close(big[:, 128:384].sum(0), host(big)[:, 128:384].astype(np.float64).sum(0), "column sums of a strided view", rt=2e-5)


@pytest.mark.gpu
def test_narrow_products_gpu(hip):
    """csrc/gemm_narrow.hip: a classifier head's products (at most 16 output columns) on the vector ALUs, against float64 --
    forward with bias, rows not a multiple of the 32 a workgroup takes, and the weight gradient `x^T @ g` accumulated
    into an existing buffer (beta = 1), with a token count that leaves a ragged last split."""
    hp = hip
    rng = np.random.default_rng(0)
    for M, K, N in ((8192, 1024, 10), (4100, 256, 16), (5000, 64, 3), (65536, 1024, 10)):
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        out = hp.empty((M, N), np.float32)
        hp.gemm(hp.from_numpy(x), hp.from_numpy(w), out, bias=hp.from_numpy(b))
        close(out, x.astype(np.float64) @ w.astype(np.float64) + b, f"narrow forward {M} x {K} x {N}")
    for T, Mc, N in ((8192, 1024, 10), (4101, 512, 16), (70000, 96, 7)):
        x = rng.standard_normal((T, Mc)).astype(np.float32)
        g = rng.standard_normal((T, N)).astype(np.float32)
        dw0 = rng.standard_normal((Mc, N)).astype(np.float32)
        dw = hp.from_numpy(dw0)
        hp.gemm(hp.from_numpy(x).T, hp.from_numpy(g), dw, beta=1.0)
        close(dw, dw0 + x.astype(np.float64).T @ g.astype(np.float64), f"narrow weight gradient {T} tokens {Mc} x {N}", rt=2e-5)


@pytest.mark.gpu
def test_mlp_full_batch_fused_equals_unfused_gpu(hip):
    """BASELINE config 2 at the benchmarked batch (65536 x 784 -> 1024 -> 1024 -> 10): one step through the Linear + ReLU
    node, the masked input gradients with their column-sum partials, the 10-column products of gemm_narrow.hip and the
    few-class cross entropy, against the same step with Linear and ReLU as the two nodes the reference builds (plain
    products, relu passes, column-sum reductions).  Loss and every parameter gradient within 1e-4 of its largest entry;
    the launch counters say which kernels each step took."""
    L, _ = _lib_hp()
    _counters(L)
    l1, g1, k1 = _mlp_step("hip:0", (784, 1024, 1024, 10), 65536, True)
    c = _counters(L)
    assert k1 == ["linear_relu", "linear_relu"] and c[16] == 2 and c[17] == 2 and c[18] == 1, (k1, c[16:19])
    l0, g0, k0 = _mlp_step("hip:0", (784, 1024, 1024, 10), 65536, False)
    c = _counters(L)
    assert k0 == ["relu", "relu"] and c[16] == 0 and c[17] == 0, (k0, c[16:18])
    assert abs(l1 - l0) <= RT * abs(l0)
    for a, b in zip(g1, g0):
        close(a, b, "parameter gradient at batch 65536, fused vs two nodes")
