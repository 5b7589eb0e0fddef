"""the reference's 1-layer Transformer example (CoLA shape) on this backend, against
vectors produced by the REAL reference running the same model definition (tests/models_transformer.py,
tools/gen_golden.py::gen_transformer).  Pins the reference-semantics LayerNorm (leading-axis statistics,
running averages used in eval), the in-place padding-mask edit under no_grad and Embedding(padding_idx)."""
import os

import numpy as np

import pydynet_amd as pdn
import pydynet_amd.nn as nn
import pydynet_amd.nn.functional as F
from pydynet_amd.core.tensor import Graph
from pydynet_amd.optim import Adam
from tests import models_transformer as mt
from tests.conftest import device_variants

G = os.path.join(os.path.dirname(__file__), "golden")
RT = 1e-4


def _host(a):
    return a if isinstance(a, np.ndarray) else a.get()


def _fused_self_attention(self, x, mask):
    """The example's SelfAttention.forward with its five score-matrix nodes replaced by ONE fused
    attention node (streaming kernels on the GPU): same projections, same in-place mask edit."""
    from pydynet_amd.core import fused
    N, L = x.shape[0], x.shape[1]
    shape = (N, L, self.heads, self.hd)
    if mask is not None:
        mask[mask.eq(1)] = np.float32('-inf')
    out = fused.attention(self.Q(x).reshape(*shape), self.K(x).reshape(*shape), self.V(x).reshape(*shape),
                          causal=False, mask=mask)
    return self.O(out.reshape(N, L, -1))


def _run(dev, fused_attention=False):
    d = np.load(os.path.join(G, "transformer_example.npz"))
    Graph.clear()
    Transformer, loss_fn = mt.build(pdn, nn, F)
    c = mt.CFG
    ids, labels, emb = mt.make_inputs()
    np.random.seed(11)
    net = Transformer(c["embed"], c["layers"], c["heads"], c["expansion"], c["vocab"], c["max_len"])
    kinds = []
    if fused_attention:
        from pydynet_amd.core import fused
        type(net.layers[0].attention).forward = _fused_self_attention      # (class is local to this build)
        orig = fused.attention.forward_

        def spy(node, *a):
            out = orig(node, *a)
            kinds.append(node._kind)
            return out
        fused.attention.forward_ = spy
    try:
        _steps(net, loss_fn, c, d, ids, labels, emb, dev)
    finally:
        if fused_attention:
            fused.attention.forward_ = orig
    if fused_attention and dev != "cpu":
        assert kinds and all(k == "stream" for k in kinds), kinds


def _steps(net, loss_fn, c, d, ids, labels, emb, dev):
    net.word_embedding.weight.data[...] = emb
    net.to(dev)
    opt = Adam(net.parameters(), lr=c["lr"])
    net.train()
    losses = []
    for s in range(c["steps"]):
        loss = loss_fn(net, pdn.Tensor(ids, device=dev), pdn.Tensor(labels, device=dev))
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.item())
        if s == 0:
            for n, p in net._parameters.items():
                if p.requires_grad:
                    g = _host(p.grad)
                    ref = float(d[f"gnorm/{n}"])
                    assert abs(float(np.linalg.norm(g.astype(np.float64))) - ref) <= RT * ref + 1e-6, n
                    if f"grad1/{n}" in d.files:            # 1e-4 relative error (north_star), norm-wise
                        r = d[f"grad1/{n}"].astype(np.float64)
                        assert np.linalg.norm(g - r) <= RT * np.linalg.norm(r) + 1e-6, n
    assert np.allclose(losses, d["losses"], rtol=RT), (losses, d["losses"])
    for n, p in net._parameters.items():
        if f"final/{n}" in d.files:                                   # running statistics of both norms
            assert np.allclose(_host(p.data), d[f"final/{n}"], rtol=RT, atol=1e-5), n
        # parameters after 3 Adam steps: an entry whose gradient is at round-off level moves by +-lr
        # in either direction (u = lr * g / (|g| + eps)), so norms are held to 1e-3 only
        ref = float(d[f"pnorm/{n}"])
        assert abs(float(np.linalg.norm(_host(p.data).astype(np.float64))) - ref) <= 1e-3 * ref + 1e-6, n
    net.eval()
    with pdn.no_grad():
        t = pdn.Tensor(ids, device=dev)
        out = net(t, pdn.unsqueeze(t.eq(0), (1, 2)).astype(np.float32))
    pdn.autograd.set_grad_enabled(True)
    assert np.allclose(_host(out.data), d["eval_out"], rtol=RT, atol=1e-5)


def test_transformer_example_cpu():
    _run("cpu")


def check_transformer_example(dev):
    _run(dev)


def check_transformer_example_fused_attention(dev):
    """Same vectors from the real reference, attention through the fused node (padding mask (B,1,1,L))."""
    _run(dev, fused_attention=True)


def test_transformer_example_fused_attention_cpu():
    _run("cpu", fused_attention=True)


device_variants(globals(), check_transformer_example)
device_variants(globals(), check_transformer_example_fused_attention)


def check_masked_attention_vs_reference(dev):
    """The attention chain of examples/pydynet/transformer.py:84-101 with its (B, 1, 1, L) padding mask, as computed by the
    REAL reference's operators (tests/golden/masked_attention.npz, tools/gen_golden_r2.py masked_attention), through
    `fused.attention`: head dim 48 / 64 on whole tiles -- on a GPU the resident kernels with the key bias (round 4) -- and
    head dim 128 at 44 positions, the example's own shape class, on csrc/attention_hd128.hip (round 6)."""
    import os
    import pydynet_amd as pdn
    from pydynet_amd.core import fused
    from pydynet_amd.core.tensor import Graph
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "masked_attention.npz"))
    for tag in ("hd48", "hd64", "hd128"):            # hd128: the example's own head dim at its own length 44 (round 6)
        Graph.clear()
        q, k, v = (pdn.Tensor(d[f"{tag}/{n}"], dtype=np.float32, device=dev, requires_grad=True) for n in "qkv")
        mask = pdn.Tensor(d[f"{tag}/pad"].copy(), dtype=np.float32, device=dev)
        with pdn.no_grad():
            mask[mask.eq(1)] = np.float32("-inf")                       # transformer.py:93
        node = fused.attention(q, k, v, causal=False, mask=mask)
        if dev != "cpu":
            assert node._kind == "resident", node._kind
        (node * pdn.Tensor(d[f"{tag}/w"], dtype=np.float32, device=dev)).sum().backward()
        for name, got in (("out", node), ("dq", q.grad), ("dk", k.grad), ("dv", v.grad)):
            got = got.numpy() if isinstance(got, pdn.Tensor) else (got if isinstance(got, np.ndarray) else got.get())
            ref = d[f"{tag}/{name}"]
            err = float(np.abs(got.astype(np.float64) - ref).max())
            assert err <= 1e-6 + 1e-4 * float(np.abs(ref).max()), (tag, name, err)


device_variants(globals(), check_masked_attention_vs_reference)


def check_example_at_its_own_width_takes_the_resident_kernels(dev):
    """examples/pydynet/transformer.py:53-130 at its OWN model width (dim 512, 4 heads = head dim 128, 44 positions,
    padding mask), written with plain operators exactly as the example writes it: on a HIP device the score chain becomes
    one fused attention node (core/fused/chain.py) and that node runs on the RESIDENT kernels (csrc/attention_hd128.hip);
    loss and every gradient equal the cpu device's plain-operator run."""
    from pydynet_amd.core import fused
    Transformer, loss_fn = mt.build(pdn, nn, F)
    rng = np.random.default_rng(5)
    B, L, V = 16, 44, 50
    ids = rng.integers(1, V, (B, L))
    for i, n in enumerate(rng.integers(10, L + 1, B)):
        ids[i, n:] = 0
    labels = rng.choice([-1.0, 1.0], B).astype(np.float32)
    emb = (0.1 * rng.standard_normal((V, 512))).astype(np.float32)
    emb[0] = 0.0
    results = {}
    for where in ("cpu", dev):
        Graph.clear()
        np.random.seed(3)
        net = Transformer(512, 1, 4, 2, V, L)
        net.word_embedding.weight.data[...] = emb
        net.to(where)
        kinds = []
        orig = fused.attention.forward_

        def spy(node, *a):
            out = orig(node, *a)
            kinds.append(node._kind)
            return out
        fused.attention.forward_ = spy
        try:
            net.train(True)
            loss = loss_fn(net, pdn.Tensor(ids, dtype=np.int64, device=where), pdn.Tensor(labels, dtype=np.float32, device=where))
            loss.backward()
        finally:
            fused.attention.forward_ = orig
        if where != "cpu":
            assert kinds == ["resident"], kinds
        results[where] = (float(loss.item()), {n: _host(p.grad).astype(np.float64) for n, p in net.named_parameters() if p.grad is not None})
    (l0, g0), (l1, g1) = results["cpu"], results[dev]
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    # the example's LayerNorm takes its statistics over the batch axes: several gradients are sums that cancel to
    # round-off (the bias in front of norm2 has NO analytic gradient) and two fp32 evaluation orders of the same step differ
    # by ~3e-4 of the largest gradient entry -- with or without the fused attention node (measured with the chain switched
    # off: identical figures).  The bar is therefore 1e-3 of the LARGEST gradient entry of the model; the attention
    # kernels themselves are held to 2e-5 in tests/test_attention_hd128.py and to the reference's vectors above.
    scale = max(float(np.abs(g).max()) for g in g0.values())
    for n in g0:
        assert float(np.abs(g0[n] - g1[n]).max()) <= 1e-3 * scale, (n, float(np.abs(g0[n] - g1[n]).max()), scale)


device_variants(globals(), check_example_at_its_own_width_takes_the_resident_kernels)
