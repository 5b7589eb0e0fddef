"""Direct (implicit-GEMM) Conv2d kernels of csrc/conv_direct.hip against a float64 NumPy statement of
nn/functional.py:254-281 (im2col + GEMM, np.add.at col2im) -- forward, data gradient, weight and
bias gradients -- over LeNet's two layers and the edge shapes the reference's conv admits (odd channel
counts, stride 2, no padding, 5x5 taps, one-pixel outputs, ragged position tails), and against the
im2col + GEMM path of the same library.  fp32 MFMA accumulates in another order than BLAS: tolerance
2e-5 relative to the tensor's largest entry (north-star bound is 1e-4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [   # N, C, H, W, O, k, stride, pad
    (5, 3, 32, 32, 20, 3, 1, 1),      # LeNet conv1 (mnist.py:82-98, 3x32x32)
    (4, 20, 16, 16, 50, 3, 1, 1),     # LeNet conv2
    (3, 1, 28, 28, 6, 5, 1, 0),       # classic LeNet-5 first layer
    (2, 5, 9, 11, 7, 3, 2, 1),        # odd channels, stride 2, ragged tail
    (2, 4, 7, 7, 33, 3, 1, 0),        # two output tiles, tiny image
    (1, 2, 3, 3, 2, 3, 1, 0),         # one output pixel
    (3, 8, 12, 12, 16, 1, 1, 0),      # 1x1 taps
    (2, 6, 10, 10, 12, 5, 1, 2),      # 5x5 same-size
]


def _ref(x, w, b, g, s, p):
    x, w, g = x.astype(np.float64), w.astype(np.float64), g.astype(np.float64)
    N, C, H, W = x.shape
    O, _, k, _ = w.shape
    xp = np.pad(x, [(0, 0), (0, 0), (p, p), (p, p)])
    oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    s0, s1, s2, s3 = xp.strides
    col = np.lib.stride_tricks.as_strided(xp, (N, C, k, k, oh, ow), (s0, s1, s2, s3, s2 * s, s3 * s))
    a = col.transpose(0, 4, 5, 1, 2, 3).reshape(N * oh * ow, -1)
    y = (a @ w.reshape(O, -1).T + (b.astype(np.float64) if b is not None else 0)).reshape(N, oh, ow, O).transpose(0, 3, 1, 2)
    g2 = g.transpose(0, 2, 3, 1).reshape(N * oh * ow, O)
    dw = (g2.T @ a).reshape(w.shape)
    db = g2.sum(0)
    dcol = (g2 @ w.reshape(O, -1)).reshape(N, oh, ow, C, k, k).transpose(0, 3, 4, 5, 1, 2)
    dxp = np.zeros_like(xp)
    t0, t1, t2, t3 = dxp.strides
    np.add.at(np.lib.stride_tricks.as_strided(dxp, (N, C, k, k, oh, ow), (t0, t1, t2, t3, t2 * s, t3 * s)), (...,), dcol)
    return y, dxp[:, :, p:p + H, p:p + W], dw, db


def _close(a, b, what):
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a.astype(np.float64) - b).max())
    assert err <= 2e-5 * scale + 1e-7, (what, err, scale)


@pytest.mark.parametrize("case", CASES)
def test_direct_conv_kernels_match_float64(hip, case):
    from pydynet_amd import _lib
    L = _lib.lib()
    N, C, H, W, O, k, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((N, C, H, W), dtype=np.float32)
    w = rng.standard_normal((O, C, k, k), dtype=np.float32)
    b = rng.standard_normal((O,), dtype=np.float32)
    oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    g = rng.standard_normal((N, O, oh, ow), dtype=np.float32)
    y_ref, dx_ref, dw_ref, db_ref = _ref(x, w, b, g, s, p)
    mask = L.query("pdn_conv2d_direct_supported", C, H, W, O, k, s, p)
    assert mask & 1 and mask & 4, (case, mask)
    X, Wd, B, G = (hip.from_numpy(a) for a in (x, w, b, g))
    Y = hip.empty((N, O, oh, ow))
    L.call("pdn_conv2d_fwd_f32", X._ptr, Wd._ptr, B._ptr, Y._ptr, N, C, H, W, O, k, s, p, hip.stream())
    _close(Y.get(), y_ref, "y")
    DW, DB = hip.empty((O, C, k, k)), hip.empty((O,))
    ws, wsb = hip.workspace(L.query("pdn_conv2d_bwd_weight_workspace_bytes", N, C, H, W, O, k, s, p))
    L.call("pdn_conv2d_bwd_weight_f32", X._ptr, G._ptr, DW._ptr, DB._ptr, 0, N, C, H, W, O, k, s, p, ws, wsb, hip.stream())
    _close(DW.get(), dw_ref, "dw")
    _close(DB.get(), db_ref, "db")
    # accumulate flag: a second call adds
    L.call("pdn_conv2d_bwd_weight_f32", X._ptr, G._ptr, DW._ptr, DB._ptr, 1, N, C, H, W, O, k, s, p, ws, wsb, hip.stream())
    _close(DW.get(), 2 * dw_ref, "dw accumulated")
    if s == 1:
        assert mask & 2
        DX = hip.empty((N, C, H, W))
        L.call("pdn_conv2d_bwd_data_f32", G._ptr, Wd._ptr, DX._ptr, N, C, H, W, O, k, s, p, hip.stream())
        _close(DX.get(), dx_ref, "dx")
    else:
        assert not mask & 2


def test_conv2d_node_direct_path_equals_im2col_path(hip):
    """The tape node on both routes (direct kernels / im2col + GEMM + col2im): same outputs and
    gradients, and the reference-layout `col` buffer is still available bit-exactly on demand."""
    import pydynet_amd as pdn
    from pydynet_amd.core import fused
    from pydynet_amd.core.tensor import Graph
    rng = np.random.default_rng(0)
    x_np = rng.standard_normal((6, 20, 16, 16), dtype=np.float32)
    w_np = rng.standard_normal((50, 20, 3, 3), dtype=np.float32)
    b_np = rng.standard_normal((1, 50, 1, 1), dtype=np.float32)
    outs = []
    for direct in (True, False):
        Graph.clear()
        fused.conv2d.use_direct = direct
        try:
            x = pdn.Tensor(x_np, dtype=np.float32, device="hip:0", requires_grad=True)
            w = pdn.Tensor(w_np, dtype=np.float32, device="hip:0", requires_grad=True)
            b = pdn.Tensor(b_np, dtype=np.float32, device="hip:0", requires_grad=True)
            node = fused.conv2d(x * 1.0, w, b, 1, 1)
            node.data                                       # (a deferred node runs its kernel at first use)
            assert bool(node._direct) == direct
            (node * node).sum().backward()
            col = node._col.get()
            outs.append([node.numpy(), x.grad.get(), w.grad.get(), b.grad.get(), col])
        finally:
            fused.conv2d.use_direct = True
    for a, b_, name in zip(outs[0], outs[1], ("y", "dx", "dw", "db", "col")):
        if name == "col":
            assert np.array_equal(a, b_)
        else:
            assert np.allclose(a, b_, rtol=1e-4, atol=1e-4 * np.abs(b_).max()), name
