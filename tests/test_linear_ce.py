"""`linear -> cross entropy` as one tape node (pydynet_amd/core/fused/dense.py: linear_cross_entropy; C ABI
pdn_linear_ce_backward_f32) against (a) the same module with the node disabled -- separate Linear and
cross-entropy nodes, the (rows, V) gradient of the logits in memory -- and (b) a float64 NumPy statement of
llm/llama/model.py:179 + nn/functional.py:364-381.  Tolerance: 1e-4 relative (north_star), gradients against
their own largest entry.  Runs on the emulated C ABI and (``-m gpu``) on a real MI355X."""
import numpy as np
import pytest

import pydynet_amd as pdn
from pydynet_amd import nn
from pydynet_amd.core import fused
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants

RT = 1e-4


def host(x):
    return x.numpy() if isinstance(x, pdn.Tensor) else (x if isinstance(x, np.ndarray) else x.get())


def close(a, b, what):
    a, b = np.asarray(host(a), np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-30)
    assert float(np.abs(a - b).max()) <= 1e-7 + RT * scale, (what, float(np.abs(a - b).max()), scale)


def _case(dev, rows, V, reduction, upstream, seed):
    D = 288
    fused.linear_cross_entropy.min_rows = 32          # (the model only takes the node from 32768 tokens up)
    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal((rows, D)).astype(np.float32)
    w0 = (0.05 * rng.standard_normal((D, V))).astype(np.float32)
    b0 = (0.1 * rng.standard_normal(V)).astype(np.float32)
    t0 = rng.integers(0, V, rows)
    t0[:3] = (0, V - 1, V // 2)
    out = {}
    for fused_on in (True, False):
        Graph.clear()
        head = nn.Linear(D, V, dtype=np.float32)
        head.weight.data[...] = w0
        head.bias.data[...] = b0
        head.to(dev)
        head.weight.zero_grad(); head.bias.zero_grad()
        x = pdn.Tensor(x0, dtype=np.float32, device=dev, requires_grad=True)
        t = pdn.Tensor(t0, dtype=np.int64, device=dev)
        h = x * 1.0                                     # a non-leaf input, as the final norm of the model is
        if fused_on:
            assert fused.linear_cross_entropy.applicable(h, head.weight, head.bias, t, reduction)
            loss = fused.linear_cross_entropy(h, head.weight, head.bias, t, reduction)
        else:
            from pydynet_amd.core.fused import chain
            chain.loss_chain.enabled = False            # the SEPARATE nodes (a pending projection would be taken over, chain.py)
            try:
                loss = nn.CrossEntropyLoss(reduction=reduction)(head(h), t)
            finally:
                chain.loss_chain.enabled = True
            assert type(loss) is not fused.linear_cross_entropy
        (loss * upstream).backward()
        out[fused_on] = (host(loss), host(x.grad), host(head.weight.grad), host(head.bias.grad))
    # float64 statement
    z = x0.astype(np.float64) @ w0.astype(np.float64) + b0
    m = z.max(-1, keepdims=True)
    lse = np.log(np.exp(z - m).sum(-1, keepdims=True)) + m
    per_row = lse[:, 0] - z[np.arange(rows), t0]
    ref_loss = per_row.mean() if reduction == "mean" else per_row.sum()
    d = np.exp(z - lse)
    d[np.arange(rows), t0] -= 1
    d *= upstream / rows if reduction == "mean" else upstream
    ref = (ref_loss, d @ w0.astype(np.float64).T, x0.astype(np.float64).T @ d, d.sum(0))
    for name, i in (("loss", 0), ("dx", 1), ("dW", 2), ("db", 3)):
        close(out[True][i], ref[i], f"fused {name} vs float64")
        close(out[True][i], out[False][i], f"fused {name} vs separate nodes")


def check_linear_ce_mean(dev):
    _case(dev, 64, 96, "mean", 1.0, 0)


def _spied(dev, attr, flags, *case):
    """Runs `_case(*case)` with the class switches `flags` and returns the node's `attr` per forward pass."""
    taken, fwd = [], fused.linear_cross_entropy.forward_
    saved = {k: getattr(fused.linear_cross_entropy, k) for k in flags}

    def spy(node, *a):
        out = fwd(node, *a)
        taken.append(getattr(node, attr))
        return out
    fused.linear_cross_entropy.forward_ = spy
    for k, v in flags.items():
        setattr(fused.linear_cross_entropy, k, v)
    try:
        _case(dev, *case)
    finally:
        fused.linear_cross_entropy.forward_ = fwd
        for k, v in saved.items():
            setattr(fused.linear_cross_entropy, k, v)
    return taken


def check_linear_ce_statistics_in_the_projection(dev):
    """Enough rows for the projection to leave the log-sum-exp itself (pdn_linear_lse_fwd_f32): the node must take it."""
    assert _spied(dev, "stats_in_gemm", {"deferred_norm": False}, 49152, 128, "mean", 1.0, 5) == [True]


def check_linear_ce_statistics_split_over_both_products(dev):
    """Row maxima from the projection, the sum of exponentials from the input-gradient product run in the forward pass
    (pdn_linear_rowmax_fwd_f32 + pdn_linear_ce_dx_deferred_f32); the upstream scalar is applied in backward."""
    assert _spied(dev, "deferred", {}, 49152, 160, "mean", 0.5, 6) == [True]
    assert _spied(dev, "deferred", {}, 57344 + 32, 96, "sum", 1.0, 7) == [True]      # 8-wave workgroups, a ragged last one
    # few rows: the projection splits its chunks, the input-gradient product its contraction, over the grid
    assert _spied(dev, "deferred", {}, 4096, 1536, "mean", 1.0, 8) == [True]


def check_linear_ce_sum_scaled_upstream(dev):
    _case(dev, 96, 160, "sum", 0.5, 1)


def check_linear_ce_many_rows_two_k_splits(dev):
    # enough rows for the weight-gradient kernel to split the tokens over the grid
    _case(dev, 4096, 256, "mean", 1.0, 2)


def check_linear_ce_few_rows_input_gradient_split_over_the_vocabulary(dev):
    # 16384 tokens = per-GPU batch 64: 128 row workgroups would leave half the chip idle, so the input-gradient
    # product cuts K = vocabulary into ranges over grid.y (pdn_gemm_outres_plan) and adds the slabs in a fixed order
    _case(dev, 16384, 3072, "mean", 1.0, 3)


def check_linear_ce_not_applicable_falls_back(dev):
    Graph.clear()
    head = nn.Linear(96, 64, dtype=np.float32)
    head.to(dev)
    x = pdn.Tensor(np.zeros((32, 96), np.float32), device=dev, requires_grad=True)
    t = pdn.Tensor(np.zeros(32, np.int64), dtype=np.int64, device=dev)
    assert not fused.linear_cross_entropy.applicable(x, head.weight, head.bias, t)       # in_features != 288
    head2 = nn.Linear(288, 64, dtype=np.float32)
    head2.to(dev)
    x2 = pdn.Tensor(np.zeros((30, 288), np.float32), device=dev, requires_grad=True)
    t2 = pdn.Tensor(np.zeros(30, np.int64), dtype=np.int64, device=dev)
    assert not fused.linear_cross_entropy.applicable(x2, head2.weight, head2.bias, t2)   # rows not a multiple of 32


for _f in (check_linear_ce_mean, check_linear_ce_statistics_in_the_projection, check_linear_ce_statistics_split_over_both_products,
           check_linear_ce_sum_scaled_upstream, check_linear_ce_many_rows_two_k_splits,
           check_linear_ce_not_applicable_falls_back):
    device_variants(globals(), _f)


@pytest.mark.gpu
def test_linear_ce_few_rows_input_gradient_split_over_the_vocabulary_gpu(hip):
    # (real kernels only: the emulated ABI has no K split to exercise, and the float64 statement is slow on CPU)
    Graph.clear()
    check_linear_ce_few_rows_input_gradient_split_over_the_vocabulary("hip:0")
    # the same with the statistics taken from the two products (vocabulary ranges: 2 in the projection, 4 in the gradient)
    assert _spied("hip:0", "deferred", {}, 16384, 3072, "mean", 0.25, 4) == [True]
    assert _spied("hip:0", "deferred", {"deferred_norm": False}, 16384, 3072, "mean", 1.0, 3) == [False]
