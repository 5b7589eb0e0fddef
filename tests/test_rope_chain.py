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


class _modes:
    """(lazy links, backward-only node) class switches for the duration of a block."""

    def __init__(self, lazy, tail):
        self.want = (lazy, tail)

    def __enter__(self):
        self.saved = (chain.rope_link.enabled, chain.rope_chain.enabled)
        chain.rope_link.enabled, chain.rope_chain.enabled = self.want

    def __exit__(self, *a):
        chain.rope_link.enabled, chain.rope_chain.enabled = self.saved


def check_rope_becomes_one_node(dev):
    """Three formulations of the same tape: every operator pending and ONE fused.rope node at the end (default); the plain
    operators run and only the backward is one node (`rope_taken`); thirteen plain nodes.  Same values (the fused kernel may
    contract r c - i s into a multiply-add: 1e-6), same gradients, float64 as the referee."""
    from pydynet_amd.core import fused
    got = {}
    for mode in ("lazy", "tail", "plain"):
        x_np, ang, w_np, x, cos, sin, w = _inputs(dev)
        built, taken = chain.rope_link.fused_built, chain.rope_chain.taken
        with _modes(mode == "lazy", mode in ("lazy", "tail")):
            out, _ = rotate(x, cos, sin)
        assert (chain.rope_link.fused_built - built, chain.rope_chain.taken - taken) == \
            {"lazy": (1, 0), "tail": (0, 1), "plain": (0, 0)}[mode]
        assert type(out) is {"lazy": fused.rope, "tail": chain.rope_taken}.get(mode, type(out))
        assert mode != "plain" or type(out) not in (fused.rope, chain.rope_taken)
        (out * w).sum().backward()
        got[mode] = (out.numpy().copy(), _host(x.grad).copy())
    assert np.array_equal(got["tail"][0], got["plain"][0])                # forward values: the plain operators' own
    c, s = np.cos(ang.astype(np.float64))[None, :, None, :], np.sin(ang.astype(np.float64))[None, :, None, :]
    want = np.empty(x_np.shape, np.float64)
    want[..., 0::2] = x_np[..., 0::2] * c - x_np[..., 1::2] * s
    want[..., 1::2] = x_np[..., 0::2] * s + x_np[..., 1::2] * c
    ref = _ref_grad(ang, w_np)
    for mode in got:
        assert np.abs(got[mode][0] - want).max() <= 2e-6 * np.abs(want).max(), mode
        assert np.abs(got[mode][1] - ref).max() <= 1e-5 * np.abs(ref).max(), mode


def check_links_that_somebody_reads(dev):
    """Reading a link at any stage runs the ordinary operators from there on; the result and its gradient do not change."""
    x_np, ang, w_np, x0, cos, sin, w = _inputs(dev, seed=4)
    ref = _ref_grad(ang, w_np)
    for stage in ("pairs", "comp", "prod", "diff", "unsq", "cat"):
        x_np, ang, w_np, x, cos, sin, w = _inputs(dev, seed=4)
        xri = x.reshape(*(x.shape[:-1] + (-1, 2)))
        if stage == "pairs":
            assert xri.numpy().shape == (B, L, H, HD // 2, 2)
        r, i = xri[..., 0], xri[..., 1]
        if stage == "comp":
            assert np.array_equal(r.numpy(), x_np[..., 0::2])
        c, s = pdn.unsqueeze(cos, axis=-2), pdn.unsqueeze(sin, axis=-2)
        rc = r * c
        if stage == "prod":
            assert rc.numpy().shape == (B, L, H, HD // 2)
        d = rc - i * s
        if stage == "diff":
            assert np.isfinite(d.numpy()).all()
        out_r, out_i = pdn.unsqueeze(d, -1), pdn.unsqueeze(r * s + i * c, -1)
        if stage == "unsq":
            assert out_r.numpy().shape == (B, L, H, HD // 2, 1)
        cat = pdn.concat([out_r, out_i], axis=-1)
        if stage == "cat":
            assert cat.numpy().shape == (B, L, H, HD // 2, 2)
        out = cat.reshape(*(cat.shape[:-2] + (-1,)))
        (out * w).sum().backward()
        assert np.abs(_host(x.grad) - ref).max() <= 1e-5 * np.abs(ref).max(), stage


def check_rope_intermediate_with_a_second_consumer(dev):
    from pydynet_amd.core import fused
    x_np, ang, w_np, x, cos, sin, w = _inputs(dev, seed=1)
    out, r = rotate(x, cos, sin)
    assert type(out) is fused.rope
    ((out * w).sum() + r.sum() * 0.5).backward()                            # r = xri[..., 0] feeds a second term
    ref = _ref_grad(ang, w_np, extra=0.5)
    assert np.abs(_host(x.grad) - ref).max() <= 1e-5 * np.abs(ref).max()


def check_other_expressions_are_left_alone(dev):
    from pydynet_amd.core import fused
    before = (chain.rope_chain.taken, chain.rope_link.fused_built)
    x_np, ang, w_np, x, cos, sin, w = _inputs(dev, seed=2)
    out, _ = rotate(x, cos, sin, flip=True)                                 # not a rotation
    assert type(out) not in (chain.rope_taken, fused.rope)
    want_r = x_np[..., 0::2] * np.cos(ang)[None, :, None, :] - x_np[..., 1::2] * np.sin(ang)[None, :, None, :]
    assert np.abs(out.numpy()[..., 0::2] - want_r).max() <= 1e-5
    x_np, ang, w_np, x, cos, sin, w = _inputs(dev, seed=2, table_grad=True)  # tables that need a gradient
    out, _ = rotate(x, cos, sin)
    assert type(out) not in (chain.rope_taken, fused.rope)
    (out * w).sum().backward()
    assert cos.grad is not None and sin.grad is not None
    x_np, ang, w_np, x, cos, sin, w = _inputs(dev, seed=2)
    cat = pdn.concat([x, x], axis=-1)                                        # some other concat, reshaped
    assert cat.reshape(B, L, H * 2 * HD).shape == (B, L, H * 2 * HD)
    with pdn.no_grad():
        out, _ = rotate(x, cos, sin)                                         # no tape: nothing to take
    assert type(out) is not chain.rope_taken
    assert (chain.rope_chain.taken, chain.rope_link.fused_built - 1) == before   # (without a tape the lazy form still fuses: one kernel)


device_variants(globals(), check_rope_becomes_one_node)
device_variants(globals(), check_links_that_somebody_reads)
device_variants(globals(), check_rope_intermediate_with_a_second_consumer)
device_variants(globals(), check_other_expressions_are_left_alone)
