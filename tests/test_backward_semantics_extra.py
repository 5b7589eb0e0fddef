"""Engine behaviours the reference has and round 1 broke: an op node whose own graph was
already walked and freed is a plain leaf of the next graph (hidden state carried across batches)."""
import numpy as np
import pytest

import pydynet_amd as pdn
import pydynet_amd.nn as nn
import pydynet_amd.nn.functional as F
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants


def check_freed_intermediate_is_a_leaf_of_the_next_graph(dev):
    Graph.clear()
    w = pdn.Tensor(np.array([1.0, 2.0, 3.0], np.float32), dtype=np.float32, device=dev, requires_grad=True)
    h = w * 2
    (h * h).sum().backward()
    g1 = w.grad.copy() if isinstance(w.grad, np.ndarray) else w.grad.get()
    assert np.allclose(g1, [8, 16, 24])
    (h * 3).sum().backward()                         # h's graph is gone: the gradient stops at h
    hg = h.grad if isinstance(h.grad, np.ndarray) else h.grad.get()
    assert np.allclose(hg, [3, 3, 3])
    g2 = w.grad if isinstance(w.grad, np.ndarray) else w.grad.get()
    assert np.allclose(g2, g1)


device_variants(globals(), check_freed_intermediate_is_a_leaf_of_the_next_graph)


def test_freed_intermediate_cpu():
    check_freed_intermediate_is_a_leaf_of_the_next_graph("cpu")


def check_float64_operands_never_reach_float32_kernels(dev):
    """nn.Linear without dtype= is float64 (as in the reference): on a HIP device it computes correctly
    through the generic kernels (float64 MFMA matmul) -- the float32 fused kernels never see the buffers;
    shapes without a float64 kernel (conv) refuse loudly."""
    Graph.clear()
    np.random.seed(0)
    lin = nn.Linear(8, 4).to(dev)
    x = pdn.Tensor(np.random.randn(5, 8), device=dev)
    ref = x.numpy() @ lin.weight.numpy() + lin.bias.numpy()
    y = lin(x)                                        # generic float64 path: pdn_gemm_f64 + broadcast add
    assert y.dtype == np.float64 and np.allclose(y.numpy(), ref, rtol=1e-12)
    x.requires_grad = True
    with pytest.raises(TypeError):
        F.conv2d(pdn.Tensor(np.zeros((1, 1, 4, 4)), device=dev), pdn.Tensor(np.zeros((1, 1, 3, 3)), device=dev))


device_variants(globals(), check_float64_operands_never_reach_float32_kernels)


def test_every_optimizer_applies_grad_scale():
    from pydynet_amd import optim
    for cls, kw in ((optim.SGD, dict(lr=0.1)), (optim.Adagrad, {}), (optim.Adadelta, {}), (optim.Adam, {})):
        outs = []
        for scale, mult in ((1.0, 1.0), (0.5, 2.0)):
            Graph.clear()
            p = pdn.Tensor(np.array([1.0, -2.0, 3.0]), dtype=np.float64, requires_grad=True)
            opt = cls([p], **kw)
            opt.grad_scale = scale
            p.grad[...] = mult * np.array([0.3, -0.1, 0.2])
            opt.step()
            outs.append(p.data.copy())
        assert np.allclose(outs[0], outs[1], rtol=1e-12), cls.__name__


def test_stale_node_with_two_consumers_never_mutates_an_aliased_gradient():
    """A freed intermediate reused in a later graph is a leaf of that graph.  Its first gradient contribution is
    adopted without a copy and may be ANOTHER node's gradient array (pass-through ops such as add hand their
    upstream gradient on unchanged); the second contribution must not be added into that array in place.
    The reference zero-initialises every grad and accumulates (tensor.py:90, 371), so its answer is the plain sum."""
    import pydynet_amd as pdn
    x = pdn.Tensor(np.array([1.0, 2.0, 3.0]), requires_grad=True)
    w = pdn.Tensor(np.array([2.0, 2.0, 2.0]), requires_grad=True)
    c = pdn.Tensor(np.array([1.0, 2.0, 3.0]))
    h0 = pdn.Tensor(np.array([1.0, 1.0, 1.0]), requires_grad=True)
    h = h0 * 2
    h.sum().backward()                          # h's graph is walked and freed: h is stale now
    xx = x * 3
    z = h * w
    y = h + xx                                  # add: passes its upstream gradient to BOTH inputs unchanged
    ((y * c).sum() + (z * c).sum()).backward()
    assert np.allclose(x.grad, 3 * c.data)                              # d/dx = 3 c, not polluted by h's share
    assert np.allclose(h.grad, c.data + c.data * w.data)
    assert np.allclose(w.grad, h.data * c.data)
    # the variant whose adopted gradient is the read-only broadcast of ones
    x.zero_grad(); w.zero_grad()
    h = h0 * 2
    h.sum().backward()
    xx = x * 3
    z = h * w
    y = h + xx
    (y.sum() + z.sum()).backward()
    assert np.allclose(x.grad, 3.0) and np.allclose(h.grad, 1.0 + w.data)
