"""The loss of the reference's own training step built from PLAIN operators (llm/llama/model.py:239-249):

    logits = lm_head(h)                                   # nn.Linear(dim, vocab)
    loss = nn.CrossEntropyLoss()(logits.reshape(B * L, V), targets)

On a HIP device the projection is a pending node; `reshape` keeps it pending and the loss takes it over as ONE
`linear_cross_entropy` node (core/fused/chain.py).  Checked here on the emulated device and on the GPU: the fused node is
built, loss and gradients equal the three-node formulation and float64, and `logits` stays usable (read, or used in a second
differentiable term) after the loss took the projection over."""
import numpy as np

import pydynet_amd as pdn
import pydynet_amd.nn as nn
from pydynet_amd.core import fused
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants

B, L, D, V = 2, 32, 288, 320


def _setup(dev, seed=3):
    rng = np.random.default_rng(seed)
    h_np = rng.standard_normal((B, L, D)).astype(np.float32)
    tgt = rng.integers(0, V, B * L)
    Graph.clear()
    np.random.seed(seed)
    head = nn.Linear(D, V, dtype=np.float32).to(dev)
    h = pdn.Tensor(h_np, device=dev, requires_grad=True)
    return h_np, tgt, head, h


def _reference(h_np, tgt, head, extra=0.0):
    w = np.asarray(head.weight.numpy(), np.float64)
    b = np.asarray(head.bias.numpy(), np.float64).reshape(-1)
    x = h_np.reshape(-1, D).astype(np.float64)
    z = x @ w + b
    m = z.max(1, keepdims=True)
    lse = m + np.log(np.exp(z - m).sum(1, keepdims=True))
    loss = float((lse[:, 0] - z[np.arange(len(tgt)), tgt]).mean()) + extra * float(z.sum())
    p = np.exp(z - lse)
    p[np.arange(len(tgt)), tgt] -= 1.0
    dz = p / len(tgt) + extra
    return loss, (dz @ w.T).reshape(h_np.shape), x.T @ dz, dz.sum(0)


class _switches:
    def __enter__(self):
        self.saved = (fused.linear_relu.min_rows, fused.linear_cross_entropy.min_rows)
        fused.linear_relu.min_rows, fused.linear_cross_entropy.min_rows = 1, 32
        return self

    def __exit__(self, *a):
        fused.linear_relu.min_rows, fused.linear_cross_entropy.min_rows = self.saved


def _close(a, b, rt=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= rt * max(np.abs(b).max(), 1e-30)


def check_plain_loss_becomes_one_node(dev):
    from pydynet_amd.core.fused import chain
    results = {}
    with _switches():
        for on in (True, False):
            h_np, tgt, head, h = _setup(dev)
            chain.loss_chain.enabled = on
            built = chain.loss_chain.fused_built
            try:
                logits = head(h)
                Bq, Lq, Vq = logits.shape                                  # (answered without running the product)
                flat = logits.reshape(Bq * Lq, Vq)
                loss = nn.CrossEntropyLoss()(flat, pdn.Tensor(tgt, dtype=np.int64, device=dev))
            finally:
                chain.loss_chain.enabled = True
            assert (chain.loss_chain.fused_built - built == 1) is on
            assert (type(loss) is fused.linear_cross_entropy) is on
            if on:
                assert logits._pending is not None and flat._pending is not None      # nobody ran the plain products
            loss.backward()
            results[on] = (loss.item(), h.grad.numpy() if hasattr(h.grad, "numpy") else np.asarray(h.grad.get()),
                           head.weight.grad, head.bias.grad)
            if on:
                ref = _reference(h_np, tgt, head)
    host = lambda g: np.asarray(g.get()) if hasattr(g, "get") else np.asarray(g)
    for on in (True, False):
        loss, dh, dw, db = results[on]
        assert abs(loss - ref[0]) <= 1e-5 * abs(ref[0])
        assert _close(host(dh), ref[1]) and _close(host(dw), ref[2]) and _close(host(db).reshape(-1), ref[3])


def check_logits_stay_usable_after_the_loss_took_the_projection(dev):
    from pydynet_amd.core.fused import chain
    with _switches():
        h_np, tgt, head, h = _setup(dev, seed=5)
        logits = head(h)
        loss = nn.CrossEntropyLoss()(logits.reshape(-1, V), pdn.Tensor(tgt, dtype=np.int64, device=dev))
        assert type(loss) is fused.linear_cross_entropy
        # a second differentiable term over the SAME logits: the passed-over projection runs now, on the tape
        total = loss + logits.sum() * 1e-3
        total.backward()
        ref = _reference(h_np, tgt, head, extra=1e-3)
        host = lambda g: np.asarray(g.get()) if hasattr(g, "get") else np.asarray(g)
        assert abs(total.item() - ref[0]) <= 1e-4 * abs(ref[0])
        assert _close(host(h.grad.data if hasattr(h.grad, "data") else h.grad), ref[1])
        assert _close(host(head.weight.grad), ref[2]) and _close(host(head.bias.grad).reshape(-1), ref[3])
        z = h_np.reshape(-1, D).astype(np.float64) @ np.asarray(head.weight.numpy(), np.float64) + \
            np.asarray(head.bias.numpy(), np.float64).reshape(-1)
        assert _close(logits.numpy().reshape(-1, V), z, 1e-5)


def check_other_reshapes_and_consumers_are_untouched(dev):
    with _switches():
        h_np, tgt, head, h = _setup(dev, seed=7)
        logits = head(h)
        r = logits.reshape(B, L * V)                                          # last axis changes: the ordinary view
        assert type(r) is not fused.linear and r.shape == (B, L * V)
        z = h_np.reshape(-1, D) @ head.weight.numpy() + head.bias.numpy().reshape(-1)
        assert _close(r.numpy().reshape(-1, V), z, 1e-5)
        flat = head(h).reshape(-1, V)                                         # regrouped, then an ordinary consumer
        assert _close((flat * 2.0).numpy(), 2.0 * z, 1e-5)
        # soft targets: the generic chain of nn/functional.py, not the fused node
        onehot = np.eye(V, dtype=np.float32)[tgt]
        soft = nn.CrossEntropyLoss()(head(h).reshape(-1, V), pdn.Tensor(onehot, device=dev))
        assert type(soft) is not fused.linear_cross_entropy and np.isfinite(soft.item())


device_variants(globals(), check_plain_loss_becomes_one_node)
device_variants(globals(), check_logits_stay_usable_after_the_loss_took_the_projection)
device_variants(globals(), check_other_reshapes_and_consumers_are_untouched)
