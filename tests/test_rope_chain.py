"""The rotary embedding written with PLAIN operators (llm/llama/model.py:23-44: reshape to pairs, two slices, four
broadcast products, a difference and a sum, two unsqueezes, concat, reshape): the final reshape puts ONE node on the tape
whose gradient is the rotation by -theta (core/fused/chain.py: rope_chain); forward values stay exactly what the plain
operators produced.  Emulated device and GPU: gradients equal the thirteen-node tape and float64; expressions that are not
this one, or whose tables need a gradient, are left alone; an intermediate with a second consumer still gets its gradient."""
import numpy as np

import pydynet_amd as pdn
from pydynet_amd.core.fused import chain
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants

B, L, H, HD = 2, 8, 3, 16


def rotate(x, cos, sin, flip=False):
    """The reference's formulation for one tensor (own restatement of model.py:23-44)."""
    xri = x.reshape(*(x.shape[:-1] + (-1, 2)))
    r, i = xri[..., 0], xri[..., 1]
    c, s = pdn.unsqueeze(cos, axis=-2), pdn.unsqueeze(sin, axis=-2)
    out_r = pdn.unsqueeze(r * c - i * s, -1)
    out_i = pdn.unsqueeze((r * s - i * c) if flip else (r * s + i * c), -1)
    out = pdn.concat([out_r, out_i], axis=-1)
    return out.reshape(*(out.shape[:-2] + (-1,))), r


def _inputs(dev, seed=0, table_grad=False):
    rng = np.random.default_rng(seed)
    x_np = rng.standard_normal((B, L, H, HD)).astype(np.float32)
    ang = rng.standard_normal((L, HD // 2)).astype(np.float32)
    w_np = rng.standard_normal((B, L, H, HD)).astype(np.float32)
    Graph.clear()
    x = pdn.Tensor(x_np, device=dev, requires_grad=True)
    cos = pdn.Tensor(np.cos(ang), device=dev, requires_grad=table_grad)
    sin = pdn.Tensor(np.sin(ang), device=dev, requires_grad=table_grad)
    return x_np, ang, w_np, x, cos, sin, pdn.Tensor(w_np, device=dev)


def _host(g):
    g = g.data if hasattr(g, "data") and not isinstance(g, np.ndarray) else g
    return np.asarray(g.get()) if hasattr(g, "get") else np.asarray(g)


def _ref_grad(ang, w_np, extra=0.0):
    c, s = np.cos(ang.astype(np.float64))[None, :, None, :], np.sin(ang.astype(np.float64))[None, :, None, :]
    gr, gi = w_np[..., 0::2].astype(np.float64), w_np[..., 1::2].astype(np.float64)
    dx = np.empty(w_np.shape, np.float64)
    dx[..., 0::2] = gr * c + gi * s + extra
    dx[..., 1::2] = -gr * s + gi * c
    return dx


def check_rope_backward_becomes_one_node(dev):
    got = {}
    for on in (True, False):
        x_np, ang, w_np, x, cos, sin, w = _inputs(dev)
        chain.rope_chain.enabled = on
        before = chain.rope_chain.taken
        try:
            out, _ = rotate(x, cos, sin)
        finally:
            chain.rope_chain.enabled = True
        assert (chain.rope_chain.taken - before == 1) is on
        assert (type(out) is chain.rope_taken) is on
        (out * w).sum().backward()
        got[on] = (out.numpy().copy(), _host(x.grad).copy())
    assert np.array_equal(got[True][0], got[False][0])                     # forward values: the plain operators' own
    ref = _ref_grad(ang, w_np)
    for on in (True, False):
        assert np.abs(got[on][1] - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.abs(got[True][1] - got[False][1]).max() <= 2e-6 * np.abs(ref).max()


def check_rope_intermediate_with_a_second_consumer(dev):
    x_np, ang, w_np, x, cos, sin, w = _inputs(dev, seed=1)
    out, r = rotate(x, cos, sin)
    assert type(out) is chain.rope_taken
    ((out * w).sum() + r.sum() * 0.5).backward()                            # r = xri[..., 0] feeds a second term
    ref = _ref_grad(ang, w_np, extra=0.5)
    assert np.abs(_host(x.grad) - ref).max() <= 1e-5 * np.abs(ref).max()


def check_other_expressions_are_left_alone(dev):
    before = chain.rope_chain.taken
    x_np, ang, w_np, x, cos, sin, w = _inputs(dev, seed=2)
    out, _ = rotate(x, cos, sin, flip=True)                                 # not a rotation
    assert type(out) is not chain.rope_taken
    x_np, ang, w_np, x, cos, sin, w = _inputs(dev, seed=2, table_grad=True)  # tables that need a gradient
    out, _ = rotate(x, cos, sin)
    assert type(out) is not chain.rope_taken
    (out * w).sum().backward()
    assert cos.grad is not None and sin.grad is not None
    x_np, ang, w_np, x, cos, sin, w = _inputs(dev, seed=2)
    cat = pdn.concat([x, x], axis=-1)                                        # some other concat, reshaped
    assert cat.reshape(B, L, H * 2 * HD).shape == (B, L, H * 2 * HD)
    with pdn.no_grad():
        out, _ = rotate(x, cos, sin)                                         # no tape: nothing to take
    assert type(out) is not chain.rope_taken
    assert chain.rope_chain.taken == before


device_variants(globals(), check_rope_backward_becomes_one_node)
device_variants(globals(), check_rope_intermediate_with_a_second_consumer)
device_variants(globals(), check_other_expressions_are_left_alone)
