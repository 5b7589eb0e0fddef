"""`F.silu(gate) * up` (llm/llama/model.py:56-58) written with plain operators becomes ONE swiglu node on a HIP device
(core/fused/chain.py: on_mul; `F.silu` is a pending node there).  Emulated device and GPU: values and gradients against
the two-node formulation and float64; other uses of the activation (read, broadcast product, scalar product) unchanged."""
import numpy as np

import pydynet_amd as pdn
import pydynet_amd.nn.functional as F
from pydynet_amd.core import fused
from pydynet_amd.core.fused import chain
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants


def _host(g):
    g = g.data if hasattr(g, "data") and not isinstance(g, np.ndarray) else g
    return np.asarray(g.get()) if hasattr(g, "get") else np.asarray(g)


def _ref(g, u, w):
    g, u, w = (v.astype(np.float64) for v in (g, u, w))
    s = 1 / (1 + np.exp(-g))
    return g * s * u, w * u * s * (1 + g * (1 - s)), w * g * s


def check_silu_times_up_is_one_node(dev):
    rng = np.random.default_rng(0)
    g_np, u_np, w_np = (rng.standard_normal((6, 40)).astype(np.float32) for _ in range(3))
    got = {}
    for on in (True, False):
        Graph.clear()
        g, u = pdn.Tensor(g_np, device=dev, requires_grad=True), pdn.Tensor(u_np, device=dev, requires_grad=True)
        chain.swiglu_chain.enabled = on
        before = chain.swiglu_chain.taken
        try:
            y = (F.silu(g) * u) if on else (u * F.silu(g))            # either order of the factors
            y2 = u * F.silu(g)
        finally:
            chain.swiglu_chain.enabled = True
        assert (type(y) is fused.swiglu) is on and (chain.swiglu_chain.taken - before == 2) is on
        (y * pdn.Tensor(w_np, device=dev)).sum().backward()
        got[on] = (y.numpy().copy(), y2.numpy().copy(), _host(g.grad).copy(), _host(u.grad).copy())
    ref = _ref(g_np, u_np, w_np)
    for on in (True, False):
        for a, r in zip((got[on][0], got[on][2], got[on][3]), ref):
            assert np.abs(a - r).max() <= 1e-5 * max(np.abs(r).max(), 1e-30)
        assert np.abs(got[on][1] - ref[0]).max() <= 1e-5 * np.abs(ref[0]).max()


def check_other_uses_of_the_activation(dev):
    rng = np.random.default_rng(1)
    g_np = rng.standard_normal((6, 40)).astype(np.float32)
    row = rng.standard_normal((40,)).astype(np.float32)
    Graph.clear()
    g = pdn.Tensor(g_np, device=dev, requires_grad=True)
    a = F.silu(g)
    want = g_np.astype(np.float64) / (1 + np.exp(-g_np.astype(np.float64)))
    assert a.shape == (6, 40) and np.abs(a.numpy() - want).max() <= 1e-6 * np.abs(want).max()     # read: runs the node
    b = F.silu(g) * pdn.Tensor(row, device=dev)                                                      # broadcast: plain product
    assert type(b) is not fused.swiglu and np.abs(b.numpy() - want * row).max() <= 1e-5
    c = F.silu(g) * 2.0                                                                              # host scalar
    assert type(c) is not fused.swiglu and np.abs(c.numpy() - 2 * want).max() <= 1e-5
    s = F.silu(g)
    d = s * s                                                                                        # the same node twice
    assert type(d) is not fused.swiglu and np.abs(d.numpy() - want * want).max() <= 1e-5
    (b.sum() + c.sum() + d.sum() + F.silu(g).sum()).backward()
    sg = 1 / (1 + np.exp(-g_np.astype(np.float64)))
    ds = sg * (1 + g_np * (1 - sg))
    ref = ds * (row + 2.0 + 2 * want + 1.0)
    assert np.abs(_host(g.grad) - ref).max() <= 1e-5 * np.abs(ref).max()


device_variants(globals(), check_silu_times_up_is_one_node)
device_variants(globals(), check_other_uses_of_the_activation)
