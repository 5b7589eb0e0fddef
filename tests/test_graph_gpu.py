"""hipGraph capture / replay of whole training steps (hipnp.Graph): a replayed step must be the SAME step
as the eager one -- same losses, same parameters -- for the MNIST-shaped MLP of examples/pydynet/mnist.py:65-79
(Linear / ReLU / cross entropy / Adam) and for a small Llama (fused attention, RMSNorm, SwiGLU, embedding
scatter), with the private pool keeping every buffer the graph refers to alive and no driver allocation
happening during replays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mlp():
    import pydynet_amd.nn as nn
    import pydynet_amd.nn.functional as F

    class MLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.layer1 = nn.Linear(784, 256, dtype=np.float32)
            self.layer2 = nn.Linear(256, 256, dtype=np.float32)
            self.layer3 = nn.Linear(256, 10, dtype=np.float32)

        def forward(self, x):
            x = x.reshape(x.shape[0], -1)
            return self.layer3(F.relu(self.layer2(F.relu(self.layer1(x)))))
    return MLP


def test_replayed_mlp_steps_equal_eager_steps(hip):
    import pydynet_amd as pdn
    import pydynet_amd.nn.functional as F
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    rng = np.random.default_rng(0)
    X = rng.random((64, 1, 28, 28), dtype=np.float32)
    y = rng.integers(0, 10, 64)
    N = 6
    results = []
    for use_graph in (False, True):
        Graph.clear()
        np.random.seed(3)
        net = _mlp()().to("hip:0")
        opt = Adam(net.parameters(), lr=1e-3)
        Xd = pdn.Tensor(X, dtype=np.float32, device="hip:0")
        yd = pdn.Tensor(y, dtype=np.int64, device="hip:0")

        def step():
            loss = F.cross_entropy_loss(net(Xd), yd)
            opt.zero_grad(); loss.backward(); opt.step()
            return loss
        losses = []
        if use_graph:
            g = hip.Graph()
            before = hip.memory_stats()["device_allocs"]
            loss = g.capture(step)                      # = two optimizer steps (warm-up + first replay)
            losses.append(loss.item())
            mid = hip.memory_stats()["device_allocs"]
            for _ in range(N - 2):
                g.replay()
                losses.append(loss.item())
            assert hip.memory_stats()["device_allocs"] == mid      # replays allocate nothing
            assert g.nodes > 10 and opt.t == 1 + N
            g.destroy()
        else:
            for _ in range(N):
                losses.append(step().item())
            losses = losses[1:]                         # (the graph run cannot read the warm-up step's loss)
        results.append((losses, {n: p.numpy() for n, p in net.named_parameters()}))
    (l0, p0), (l1, p1) = results
    assert np.allclose(l0, l1, rtol=1e-6), (l0, l1)
    for n in p0:
        assert np.allclose(p0[n], p1[n], rtol=1e-4, atol=2e-6), n     # (device-side double a_t vs the host scalar)


def test_replayed_llama_steps_equal_eager_steps(hip):
    import pydynet_amd as pdn
    from pydynet_amd.llm.llama import Llama
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    rng = np.random.default_rng(1)
    V, D, H, F_, L, B = 128, 96, 2, 128, 64, 2
    ids, tgt = rng.integers(0, V, (B, L)), rng.integers(0, V, (B * L,))
    N = 5
    results = []
    for use_graph in (False, True):
        Graph.clear()
        np.random.seed(7)
        m = Llama(V, D, H, F_, 64, B, 2, np.float32)
        m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(V, D)).astype(np.float32)
        m.to("hip:0")
        opt = Adam(m.parameters(), lr=1e-3)
        opt.flatten_grads()
        idd = pdn.Tensor(ids, dtype=np.int64, device="hip:0")
        tgd = pdn.Tensor(tgt, dtype=np.int64, device="hip:0")
        m.train(True)

        def step():
            opt.zero_grad()
            loss = m.loss(idd, tgd)
            loss.backward()
            opt.step()
            return loss
        if use_graph:
            g = hip.Graph()
            loss = g.capture(step)
            losses = [loss.item()]
            for _ in range(N - 2):
                g.replay()
                losses.append(loss.item())
            g.destroy()
        else:
            losses = [step().item() for _ in range(N)][1:]
        results.append((losses, {n: p.numpy() for n, p in m.named_parameters()}))
    (l0, p0), (l1, p1) = results
    assert np.allclose(l0, l1, rtol=1e-6), (l0, l1)
    for n in p0:
        assert np.allclose(p0[n], p1[n], rtol=1e-4, atol=2e-6), n     # (device-side double a_t vs the host scalar)


def test_replayed_transformer_example_steps_equal_the_reference_losses(hip):
    """examples/pydynet/transformer.py's step (plain-operator model code: embedding scatter, in-place -inf padding mask under
    no_grad, the recognised attention chain, leading-axis LayerNorm with running statistics, logistic loss, Adam) captured
    once and replayed: step 2 and step 3 against the losses the REAL reference produced (tests/golden/transformer_example.npz)
    -- what `bench.py --config transformer` times by default."""
    import os
    import pydynet_amd as pdn
    import pydynet_amd.nn as nn
    import pydynet_amd.nn.functional as F
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    from tests import models_transformer as mt
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transformer_example.npz"))
    c = mt.CFG
    Transformer, loss_fn = mt.build(pdn, nn, F)
    ids_np, labels_np, emb = mt.make_inputs()
    Graph.clear()
    np.random.seed(11)
    net = Transformer(c["embed"], c["layers"], c["heads"], c["expansion"], c["vocab"], c["max_len"])
    net.word_embedding.weight.data[...] = emb
    net.to("hip:0")
    opt = Adam(net.parameters(), lr=c["lr"])
    net.train()
    ids, labels = pdn.Tensor(ids_np, device="hip:0"), pdn.Tensor(labels_np, device="hip:0")

    def step():
        loss = loss_fn(net, ids, labels)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
    g = hip.Graph()
    loss = g.capture(step)                                  # warm-up run = step 1, first replay = step 2
    got = [loss.item()]
    g.replay()
    got.append(loss.item())
    assert g.nodes > 20 and opt.t == 4
    g.destroy()
    assert np.allclose(got, d["losses"][1:3], rtol=1e-4), (got, d["losses"])


def test_replayed_gru_steps_equal_eager_steps(hip):
    """examples/pydynet/ts_prediction.py's step (GRU over T steps on the persistent sequence kernels, Linear head, MSE, Adam)
    replayed as one hipGraph = the same steps issued eagerly (what `bench.py --config gru` times by default)."""
    import pydynet_amd as pdn
    import pydynet_amd.nn as nn
    import pydynet_amd.nn.functional as F
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    rng = np.random.default_rng(5)
    T, B, H, N = 12, 96, 32, 5
    xs_np, ys_np = rng.random((T, B, 1), dtype=np.float32), rng.random((B, 1), dtype=np.float32)
    results = []
    for use_graph in (False, True):
        Graph.clear()
        np.random.seed(2)
        gru = nn.GRU(1, H, dtype=np.float32).to("hip:0")
        head = nn.Linear(H, 1, dtype=np.float32).to("hip:0")
        params = list(gru.parameters()) + list(head.parameters())
        opt = Adam(params, lr=1e-3)
        xs, ys = pdn.Tensor(xs_np, device="hip:0"), pdn.Tensor(ys_np, device="hip:0")

        def step():
            out, hn = gru(xs)
            loss = F.mse_loss(head(hn[0]), ys)
            opt.zero_grad(); loss.backward(); opt.step()
            return loss
        if use_graph:
            g = hip.Graph()
            loss = g.capture(step)
            losses = [loss.item()]
            for _ in range(N - 2):
                g.replay()
                losses.append(loss.item())
            g.destroy()
        else:
            losses = [step().item() for _ in range(N)][1:]
        results.append((losses, [p.numpy() for p in params]))
    (l0, p0), (l1, p1) = results
    assert np.allclose(l0, l1, rtol=1e-6), (l0, l1)
    for a, b in zip(p0, p1):
        assert np.allclose(a, b, rtol=1e-4, atol=2e-6)


def test_replay_survives_a_later_eager_op_that_grows_the_workspace(hip):
    """A captured step holds raw scratch addresses (split-K slabs, reductions, the embedding scatter's
    last-occurrence vector, Adam's chunk table).  The process-wide scratch buffer is replaced whenever an eager op
    needs a bigger one; the block the graph still writes into must not go back to the allocator.  Run eagerly,
    capture, run eager ops with a much bigger scratch need and allocate arrays that would receive the freed block,
    replay, compare with an eager run of the same steps."""
    import pydynet_amd as pdn
    import pydynet_amd.nn.functional as F
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    rng = np.random.default_rng(0)
    X = rng.random((64, 1, 28, 28), dtype=np.float32)
    y = rng.integers(0, 10, 64)
    results = []
    for use_graph in (False, True):
        Graph.clear()
        np.random.seed(3)
        net = _mlp()().to("hip:0")
        opt = Adam(net.parameters(), lr=1e-3)
        Xd = pdn.Tensor(X, dtype=np.float32, device="hip:0")
        yd = pdn.Tensor(y, dtype=np.int64, device="hip:0")

        def step():
            loss = F.cross_entropy_loss(net(Xd), yd)
            opt.zero_grad(); loss.backward(); opt.step()
            return loss
        step()                                              # eager first: the process-wide scratch exists
        if use_graph:
            g = hip.Graph()
            loss = g.capture(step)
        else:
            step(); loss = step()
        # a later eager op with a far bigger scratch need (split-K weight gradient), then arrays that take
        # whatever block was released, filled with a pattern a stray scratch write would destroy
        a = hip.from_numpy(rng.standard_normal((65536, 96), dtype=np.float32))
        b = hip.from_numpy(rng.standard_normal((65536, 288), dtype=np.float32))
        big = hip.matmul(a.T, b)
        canaries = [hip.zeros((1 << 18,), np.float32) + 7.0 for _ in range(8)]
        hip.synchronize()
        if use_graph:
            g.replay(); g.replay()
        else:
            step(); loss = step()
        results.append((loss.item(), {n: p.numpy() for n, p in net.named_parameters()}))
        for c in canaries:
            assert np.all(c.get() == 7.0)
        assert np.isfinite(big.get()).all()
        if use_graph:
            g.destroy()
    (l0, p0), (l1, p1) = results
    assert np.allclose(l0, l1, rtol=1e-6)
    for n in p0:
        # Adam moves an entry whose gradient sits at round-off level by u = lr g / (|g| + eps) in a direction the
        # noise decides (device-side double a_t vs the host scalar): a few entries may differ by up to one step per
        # iteration; a stray scratch write would be orders of magnitude away (and is what the canaries catch)
        bad = np.abs(p0[n] - p1[n]) > 2e-6 + 1e-4 * np.abs(p0[n])
        assert bad.mean() <= 2e-3 and np.abs(p0[n] - p1[n]).max() <= 5 * 2e-3, (n, bad.mean(), np.abs(p0[n] - p1[n]).max())


def test_read_later_returns_the_value_without_draining_the_stream(hip):
    """hipnp.read_later: the host value of an array through a copy + event of its own.  Values equal `.get()`; a value
    requested BEFORE more work is queued is still the value of that moment (the copy is ordered behind the producers only)."""
    rng = np.random.default_rng(3)
    a_np = rng.standard_normal((37, 5)).astype(np.float32)
    a = hip.from_numpy(a_np)
    b = a * 2.0
    later = hip.read_later(b)
    b += 1.0                                         # queued after the copy: must not be seen by it
    big = hip.from_numpy(rng.standard_normal((2048, 2048)).astype(np.float32))
    for _ in range(4):
        big = big @ big * 1e-3                       # keeps the stream busy behind the event
    assert np.array_equal(later.get(), a_np * 2.0)
    assert np.array_equal(later.get(), a_np * 2.0)   # cached
    assert np.array_equal(b.get(), a_np * 2.0 + 1.0)
    s = hip.read_later((a * a).sum())
    assert abs(s.item() - float((a_np.astype(np.float64) ** 2).sum())) < 1e-3
