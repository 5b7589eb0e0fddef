"""conv2d -> relu -> max_pool2d(2, 2) as ONE node (fused.conv2d_relu_pool, csrc/conv_direct.hip EP = 1 / SRC = 1).

The reference composes the chain from three tape nodes (examples/pydynet/mnist.py:92-95; functional.py:31-32,
254-339).  Here a conv2d node of a shape the fused kernel takes is DEFERRED, relu of it too, and max_pool2d(., 2, 2)
launches one kernel; the backward expands the pooled gradient through a hit map (one bit per conv output position)
inside the loads of the data- / weight-gradient kernels.  Checked:
  * against a float64 NumPy statement of the reference's semantics, INCLUDING ties (integer-valued inputs make equal
    window maxima and exact zeros common): every tied position that passes relu'(y) = [y >= 0] gets the gradient
    (tensor.py:808-815);
  * against the unfused composition of the same library (bit-identical forward);
  * that any OTHER consumer of a deferred node materialises it and behaves as before;
  * the standalone mask expansion.
Runs on the real MI355X (-m gpu) and on the emulated C ABI."""
import numpy as np
import pytest

from tests.conftest import device_variants

CASES = [   # N, C, H, W, O     (k = 3, stride 1, pad 1)
    (5, 3, 32, 32, 20),       # LeNet conv1 (OW = 32: the window's rows are two chunks of one wave)
    (4, 20, 16, 16, 50),      # LeNet conv2 (OW = 16: a chunk holds two rows), two channel tiles
    (7, 20, 16, 16, 50),      # ... an odd batch: conv_quad.hip pairs images in a tile, the last one is alone
    (3, 6, 8, 8, 12),         # OW = 8: four rows per chunk
    (2, 4, 16, 32, 7),        # rectangular, odd channel count
]


def _ref(x, w, b, gp):
    """float64: pooled output and the gradients of sum(pooled * gp) w.r.t. x, w, b (reference tie semantics)."""
    x, w, b, gp = (a.astype(np.float64) for a in (x, w, b, gp))
    N, C, H, W = x.shape
    O = w.shape[0]
    xp = np.pad(x, [(0, 0), (0, 0), (1, 1), (1, 1)])
    s0, s1, s2, s3 = xp.strides
    col = np.lib.stride_tricks.as_strided(xp, (N, C, 3, 3, H, W), (s0, s1, s2, s3, s2, s3))
    a = col.transpose(0, 4, 5, 1, 2, 3).reshape(N * H * W, -1)
    y = (a @ w.reshape(O, -1).T + b).reshape(N, H, W, O).transpose(0, 3, 1, 2)
    r = np.maximum(0.0, y)
    win = r.reshape(N, O, H // 2, 2, W // 2, 2)
    pooled = win.max((3, 5))
    dr = ((win == pooled[:, :, :, None, :, None]) * gp[:, :, :, None, :, None]).reshape(N, O, H, W)
    dy = (r == y) * dr                                     # maximum(0., y): the gradient passes where out == y
    g2 = dy.transpose(0, 2, 3, 1).reshape(N * H * W, O)
    dw = (g2.T @ a).reshape(w.shape)
    db = g2.sum(0)
    dcol = (g2 @ w.reshape(O, -1)).reshape(N, H, W, C, 3, 3).transpose(0, 3, 4, 5, 1, 2)
    dxp = np.zeros_like(xp)
    t0, t1, t2, t3 = dxp.strides
    np.add.at(np.lib.stride_tricks.as_strided(dxp, (N, C, 3, 3, H, W), (t0, t1, t2, t3, t2, t3)), (...,), dcol)
    return pooled, dxp[:, :, 1:-1, 1:-1], dw, db


def _run(dev, x, w, b, gp, defer, x_grad=True):
    import pydynet_amd as pdn
    import pydynet_amd.nn.functional as F
    from pydynet_amd.core import fused
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    fused.conv2d.defer = defer
    try:
        X = pdn.Tensor(x, dtype=np.float32, device=dev, requires_grad=x_grad)
        Wt = pdn.Tensor(w, dtype=np.float32, device=dev, requires_grad=True)
        Bt = pdn.Tensor(b.reshape(1, -1, 1, 1), dtype=np.float32, device=dev, requires_grad=True)
        out = F.max_pool2d(F.relu(F.conv2d(X, Wt, 1, 1, Bt)), 2, 2)
        kind = type(out).__name__
        (out * pdn.Tensor(gp, dtype=np.float32, device=dev)).sum().backward()
        return kind, out.numpy(), (X.grad.get() if x_grad else None), Wt.grad.get(), Bt.grad.get().reshape(-1)
    finally:
        fused.conv2d.defer = True


def _quad_counters(reset=False):
    import ctypes
    from pydynet_amd import _lib
    buf = (ctypes.c_int64 * 24)()
    _lib.lib().call("pdn_kernel_counters", buf, 24, 1 if reset else 0)
    return tuple(int(v) for v in buf[21:24])


def _close(a, b, what, tol=2e-5):
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(np.asarray(a, np.float64) - b).max())
    assert err <= tol * scale + 1e-7, (what, err, scale)


def check_fused_chain_matches_float64_and_unfused(dev):
    for case in CASES:
        N, C, H, W, O = case
        rng = np.random.default_rng(sum(case))
        for integer in (False, True):
            if integer:        # small integers: exact arithmetic, many ties and exact zeros at the relu
                x = rng.integers(-2, 3, (N, C, H, W)).astype(np.float32)
                w = rng.integers(-1, 2, (O, C, 3, 3)).astype(np.float32)
                b = rng.integers(-1, 2, (O,)).astype(np.float32)
            else:
                x = rng.standard_normal((N, C, H, W), dtype=np.float32)
                w = (0.3 * rng.standard_normal((O, C, 3, 3))).astype(np.float32)
                b = rng.standard_normal((O,), dtype=np.float32)
            gp = rng.standard_normal((N, O, H // 2, W // 2), dtype=np.float32)
            _quad_counters(reset=True)
            kind, out, dx, dw, db = _run(dev, x, w, b, gp, defer=True)
            assert kind == "conv2d_relu_pool", (case, kind)
            # the LeNet shapes run on csrc/conv_quad.hip (library launch counters 21 / 22 / 23).  conv1 has no fused data
            # gradient (it is a network's first layer): asked for dx, its node expands the pooled gradient and runs the plain kernels
            want = {(3, 32, 32, 20): (1, 0, 0), (20, 16, 16, 50): (1, 1, 1)}.get((C, H, W, O), (0, 0, 0))
            assert _quad_counters() == want, (case, _quad_counters(), want)
            kind0, out0, dx0, dw0, db0 = _run(dev, x, w, b, gp, defer=False)
            assert kind0 == "pool2d"
            # the LeNet shapes run the fused forward on csrc/conv_quad.hip, whose contraction order differs from the plain
            # kernel's: identical where the arithmetic is exact (integers), fp32 round-off otherwise
            if integer:
                assert np.array_equal(out, out0), case
            else:
                _close(out, out0.astype(np.float64), (case, "vs unfused", "pooled"), 2e-6)
            ref = _ref(x, w, b, gp)
            tol = 1e-6 if integer else 2e-5
            for got, want, name in zip((out, dx, dw, db), ref, ("pooled", "dx", "dw", "db")):
                _close(got, want, (case, integer, name), tol)
            for got, want, name in zip((dx, dw, db), (dx0, dw0, db0), ("dx", "dw", "db")):
                _close(got, want.astype(np.float64), (case, "vs unfused", name), 2e-5)
        # the first layer of a network: no gradient for the input
        _quad_counters(reset=True)
        kind, out, dx, dw, db = _run(dev, x, w, b, gp, defer=True, x_grad=False)
        assert kind == "conv2d_relu_pool" and dx is None
        if (C, H, W, O) in ((3, 32, 32, 20), (20, 16, 16, 50)):
            assert _quad_counters() == (1, 0, 1), (case, _quad_counters())
        _close(dw, ref[2], (case, "dw, no dx"), 1e-6)


def check_deferred_nodes_materialise_for_any_other_consumer(dev):
    import pydynet_amd as pdn
    import pydynet_amd.nn.functional as F
    from pydynet_amd.core import fused
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 32, 32), dtype=np.float32)
    w = (0.3 * rng.standard_normal((20, 3, 3, 3))).astype(np.float32)
    X = pdn.Tensor(x, dtype=np.float32, device=dev, requires_grad=True)
    Wt = pdn.Tensor(w, dtype=np.float32, device=dev, requires_grad=True)
    c = F.conv2d(X, Wt, 1, 1)
    assert isinstance(c, fused.conv2d) and c._pending is not None          # nothing has run yet
    assert c.shape == (2, 20, 32, 32) and c.dtype == np.float32 and c.ndim == 4 and c.size == 2 * 20 * 32 * 32
    assert c._pending is not None                                          # metadata did not materialise it
    r = F.relu(c)
    assert r._pending is not None and r.shape == c.shape
    a = F.avg_pool2d(r, 2, 2)                                              # NOT the fused pattern
    assert c._pending is None and r._pending is None and type(a).__name__ == "pool2d"
    fused.conv2d.defer = False
    try:
        Graph.clear()
        X2 = pdn.Tensor(x, dtype=np.float32, device=dev, requires_grad=True)
        W2 = pdn.Tensor(w, dtype=np.float32, device=dev, requires_grad=True)
        a2 = F.avg_pool2d(F.relu(F.conv2d(X2, W2, 1, 1)), 2, 2)
    finally:
        fused.conv2d.defer = True
    assert np.array_equal(a.numpy(), a2.numpy())
    a.sum().backward(); a2.sum().backward()
    assert np.allclose(X.grad.get(), X2.grad.get(), rtol=1e-6, atol=1e-7)
    assert np.allclose(Wt.grad.get(), W2.grad.get(), rtol=1e-5, atol=1e-6)
    # a deferred conv read directly, and one whose relu is pooled with another window
    Graph.clear()
    c = F.conv2d(pdn.Tensor(x, dtype=np.float32, device=dev), pdn.Tensor(w, dtype=np.float32, device=dev), 1, 1)
    assert c._pending is not None and c.numpy().shape == (2, 20, 32, 32) and c._pending is None
    p3 = F.max_pool2d(F.relu(F.conv2d(pdn.Tensor(x, dtype=np.float32, device=dev),
                                      pdn.Tensor(w, dtype=np.float32, device=dev), 1, 1)), 4, 4)
    assert type(p3).__name__ == "pool2d" and p3.shape == (2, 20, 8, 8)


def check_fused_chain_without_bias_and_without_autograd(dev):
    """No bias operand; and under no_grad (inference) the chain is still one kernel and builds no tape."""
    import pydynet_amd as pdn
    import pydynet_amd.nn.functional as F
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3, 20, 16, 16), dtype=np.float32)
    w = (0.2 * rng.standard_normal((50, 20, 3, 3))).astype(np.float32)
    gp = rng.standard_normal((3, 50, 8, 8), dtype=np.float32)
    X = pdn.Tensor(x, dtype=np.float32, device=dev, requires_grad=True)
    Wt = pdn.Tensor(w, dtype=np.float32, device=dev, requires_grad=True)
    out = F.max_pool2d(F.relu(F.conv2d(X, Wt, 1, 1)), 2, 2)
    assert type(out).__name__ == "conv2d_relu_pool" and len(out.last) == 2
    (out * pdn.Tensor(gp, dtype=np.float32, device=dev)).sum().backward()
    ref = _ref(x, w, np.zeros(50, np.float32), gp)
    _close(out.numpy(), ref[0], "pooled (no bias)")
    _close(X.grad.get(), ref[1], "dx (no bias)")
    _close(Wt.grad.get(), ref[2], "dw (no bias)")
    with pdn.no_grad():
        n0 = Graph.size()
        y = F.max_pool2d(F.relu(F.conv2d(pdn.Tensor(x, dtype=np.float32, device=dev), pdn.Tensor(w, dtype=np.float32, device=dev),
                                         1, 1)), 2, 2)
        assert type(y).__name__ == "conv2d_relu_pool" and not y.requires_grad and Graph.size() == n0
        assert np.array_equal(y.numpy(), out.numpy())


def check_mask_expansion_entry(dev):
    from pydynet_amd import hipnp as hp, _lib
    import pydynet_amd as pdn  # noqa: F401
    from pydynet_amd.cuda import Device
    L = _lib.lib()
    rng = np.random.default_rng(3)
    rows, OH, OW = 8, 6, 12                                # rows * OH * OW = 576 = 18 hit words
    dp = rng.standard_normal((rows, OH // 2, OW // 2), dtype=np.float32)
    bits = rng.integers(0, 2, rows * OH * OW).astype(np.uint32)
    words = (bits.reshape(-1, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    with Device(dev):
        D, M = hp.from_numpy(dp), hp.from_numpy(words.view(np.int32))
        out = hp.empty((rows, OH, OW), np.float32)
        L.call("pdn_pool_mask_expand_f32", D._ptr, M._ptr, out._ptr, rows, OH, OW, hp.stream())
        got = out.get()
    want = np.where(bits.reshape(rows, OH, OW).astype(bool), np.repeat(np.repeat(dp, 2, 1), 2, 2), 0)
    assert np.array_equal(got, want)


for _fn in (check_fused_chain_matches_float64_and_unfused, check_deferred_nodes_materialise_for_any_other_consumer,
            check_fused_chain_without_bias_and_without_autograd, check_mask_expansion_entry):
    device_variants(globals(), _fn)
