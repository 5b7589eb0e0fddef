"""Generate golden vectors by running the REAL reference (imported from /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  Output:
small `.npz` / `.json` fixtures under tests/golden/ holding inputs and the reference's outputs
and gradients, keyed by our own names.  Nothing of the reference's source is stored.

    python tools/gen_golden.py
"""
import json
import os
import random
import sys
import warnings

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

import pydynet as pdn                      # noqa: E402
import pydynet.nn as nn                    # noqa: E402
import pydynet.nn.functional as F          # noqa: E402
from pydynet.optim import Adam             # noqa: E402
from pydynet.core.tensor import Graph      # noqa: E402


def fresh():
    Graph.node_list.clear(); Graph.size = 0
    pdn.autograd.set_grad_enabled(True)


def T(a, rg=False):
    return pdn.Tensor(np.array(a), dtype=np.asarray(a).dtype, requires_grad=rg)


# ---------------------------------------------------------------------------------------
def gen_ops():
    """The seeded generators of the reference's own tests (tests/test_tensor_basic.py:49-92,107)."""
    d = {}
    sys.path.insert(0, os.path.join(REF, "tests"))
    random.seed(0)
    import importlib
    tb = importlib.import_module("test_tensor_basic")
    random.seed(0)
    pairs = list(tb.array_pair_generator(tb.broadcastable_shape_pair, 4, 5, 8, seed=42))
    names = ["add", "sub", "mul", "div", "pow", "maximum", "minimum"]
    with np.errstate(all="ignore"):
        for i, (a, b) in enumerate(pairs):
            d[f"bin{i}_a"], d[f"bin{i}_b"] = a, b
            for n in names:
                fresh()
                d[f"bin{i}_{n}"] = getattr(pdn, n)(pdn.Tensor(a), pdn.Tensor(b)).data
    mm = list(tb.array_pair_generator(tb.matmul_shape_pair, 4, 5, 8, seed=42))
    for i, (a, b) in enumerate(mm):
        fresh()
        d[f"mm{i}_a"], d[f"mm{i}_b"] = a, b
        d[f"mm{i}_out"] = pdn.matmul(pdn.Tensor(a), pdn.Tensor(b)).data
    # backward cases (tests/test_backward.py) + a few float32 grads through every differentiable op
    rng = np.random.default_rng(7)
    x = rng.standard_normal((3, 4)).astype(np.float32) + 2.5
    y = rng.standard_normal((3, 4)).astype(np.float32) + 2.5
    d["g_x"], d["g_y"] = x, y
    for n in ["add", "sub", "mul", "div", "pow", "maximum"]:
        fresh()
        a, b = T(x, True), T(y, True)
        out = getattr(pdn, n)(a, b)
        (out * out).sum().backward()
        d[f"g_{n}_out"], d[f"g_{n}_da"], d[f"g_{n}_db"] = out.data, a.grad, b.grad
    for n in ["exp", "log", "sigmoid", "tanh", "sqrt", "square"]:
        fresh()
        a = T(x, True)
        out = getattr(pdn, n)(a)
        (out * out).sum().backward()
        d[f"g_{n}_out"], d[f"g_{n}_da"] = out.data, a.grad
    for n, ax, kd in [("sum", 1, False), ("mean", (0, 1), True), ("max", 0, False), ("min", None, False), ("mean", -1, True)]:
        fresh()
        a = T(x, True)
        out = getattr(pdn, n)(a, ax, kd)
        (out * out).sum().backward()
        key = f"g_{n}_{str(ax).replace(' ', '')}_{int(kd)}"
        d[key + "_out"], d[key + "_da"] = out.data, a.grad
    fresh()
    a, b = T(x, True), T(y[:, :2].copy(), True)
    out = pdn.concat([a, b], axis=1).reshape(3, 2, 3).transpose(1, 0, 2).swapaxes(0, 2)
    (out * out).sum().backward()
    d["g_views_out"], d["g_views_da"], d["g_views_db"] = out.data, a.grad, b.grad
    fresh()
    a = T(x, True); w = T(rng.standard_normal((4, 5)).astype(np.float32), True)
    (a @ w).sum().backward()
    d["g_mm_w"], d["g_mm_da"], d["g_mm_dw"] = w.data, a.grad, w.grad
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **d)


def gen_functional():
    d = {}
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((4, 6, 10)) * 2).astype(np.float32)
    d["sm_x"] = x
    for ax in (-1, None, 1):
        fresh()
        a = T(x, True)
        out = F.softmax(a, ax)
        (out * T(np.arange(out.size, dtype=np.float32).reshape(out.shape) / out.size)).sum().backward()
        d[f"sm_{ax}_out"], d[f"sm_{ax}_dx"] = out.data, a.grad
    fresh()
    a = T(x, True)
    out = F.log_softmax(a, -1, True)
    (out * out).sum().backward()
    d["lsm_out"], d["lsm_dx"] = out.data, a.grad
    # cross entropy: int targets and one-hot targets (mean over N*C quirk)
    lg = (rng.standard_normal((12, 7)) * 3).astype(np.float32)
    tg = rng.integers(0, 7, 12)
    d["ce_logits"], d["ce_t"] = lg, tg
    for red in ("mean", "sum"):
        fresh()
        a = T(lg, True)
        loss = F.cross_entropy_loss(a, pdn.Tensor(tg, dtype=np.int64), red)
        loss.backward()
        d[f"ce_{red}_loss"], d[f"ce_{red}_dx"] = loss.data, a.grad
    fresh()
    a = T(lg, True)
    loss = F.cross_entropy_loss(a, T(np.eye(7, dtype=np.float32)[tg]))
    loss.backward()
    d["ce_onehot_loss"], d["ce_onehot_dx"] = loss.data, a.grad
    # embedding with duplicate ids: scatter-ASSIGN gradient
    w = rng.standard_normal((9, 4)).astype(np.float32)
    ids = np.array([[1, 1, 3], [8, 3, 0]])
    fresh()
    W = T(w, True)
    e = F.embedding(ids, W, None)
    (e * T(np.arange(e.size, dtype=np.float32).reshape(e.shape))).sum().backward()
    d["emb_w"], d["emb_ids"], d["emb_out"], d["emb_dw"] = w, ids, e.data, W.grad
    # relu at zero, leaky relu, silu
    r = np.array([-1.5, 0.0, 0.0, 2.0, -0.0, 3.5], np.float32)
    for n, f in [("relu", F.relu), ("lrelu", lambda t: F.leaky_relu(t, 0.1)), ("silu", F.silu),
                 ("sigmoid", F.sigmoid), ("tanh", F.tanh)]:
        fresh()
        a = T(r, True)
        out = f(a)
        (out * 2.0).sum().backward()
        d[f"{n}_x"], d[f"{n}_out"], d[f"{n}_dx"] = r, out.data, a.grad
    # conv2d / pools (+ the raw im2col array for bit-exactness)
    cx = rng.standard_normal((2, 3, 8, 8)).astype(np.float32)
    ck = rng.standard_normal((5, 3, 3, 3)).astype(np.float32)
    d["conv_x"], d["conv_k"] = cx, ck
    im2col = getattr(F, "_" + "_im2col2d")
    pad2d = getattr(F, "_" + "_pad2d")
    for s, p in [(1, 0), (1, 1), (2, 1), (2, 0)]:
        fresh()
        a, k = T(cx, True), T(ck, True)
        out = F.conv2d(a, k, p, s)
        (out * out).sum().backward()
        d[f"conv_s{s}p{p}_out"], d[f"conv_s{s}p{p}_dx"], d[f"conv_s{s}p{p}_dk"] = np.ascontiguousarray(out.data), a.grad, k.grad
        fresh()
        d[f"col_s{s}p{p}"] = im2col(pad2d(T(cx), p), 3, s).data
    for n, f in [("maxpool", F.max_pool2d), ("avgpool", F.avg_pool2d)]:
        fresh()
        a = T(cx, True)
        out = f(a, 2, 2)
        (out * out).sum().backward()
        d[f"{n}_out"], d[f"{n}_dx"] = np.ascontiguousarray(out.data), a.grad
    # max pool with ties: gradient to ALL tied positions
    tie = np.zeros((1, 1, 4, 4), np.float32); tie[0, 0, :2, :2] = 1.0
    fresh()
    a = T(tie, True)
    F.max_pool2d(a, 2, 2).sum().backward()
    d["maxpool_tie_x"], d["maxpool_tie_dx"] = tie, a.grad
    # RMSNorm / LayerNorm (reference semantics: leading-axis stats + running stats)
    nx = rng.standard_normal((3, 5, 16)).astype(np.float32)
    d["norm_x"] = nx
    np.random.seed(0)
    fresh()
    rn = nn.RMSNorm(16, dtype=np.float32)
    rn.weight.data[...] = rng.standard_normal(16).astype(np.float32)
    a = T(nx, True)
    out = rn(a)
    (out * out).sum().backward()
    d["rms_w"], d["rms_out"], d["rms_dx"], d["rms_dw"] = rn.weight.data.copy(), out.data, a.grad, rn.weight.grad.copy()
    fresh()
    ln = nn.LayerNorm(16, dtype=np.float32)
    a = T(nx, True)
    o1 = ln(a)
    (o1 * o1).sum().backward()
    o2 = ln(T(nx * 2))
    d["ln_out1"], d["ln_dx"], d["ln_dscale"], d["ln_dshift"] = o1.data, a.grad, ln.scale.grad.copy(), ln.shift.grad.copy()
    d["ln_running_mean"], d["ln_running_var"] = ln.running_mean.data.copy(), ln.running_var.data.copy()
    ln.set_module_state(False)
    d["ln_eval_out"] = ln(T(nx)).data
    # recurrent cells
    np.random.seed(3)
    fresh()
    cell = nn.GRUCell(6, 8, dtype=np.float32)
    gx, gh = rng.standard_normal((5, 6)).astype(np.float32), rng.standard_normal((5, 8)).astype(np.float32)
    a, h = T(gx, True), T(gh, True)
    out = cell(a, h)
    (out * out).sum().backward()
    for n in ["Wx1", "Wx2", "Wh1", "Wh2", "bias1", "bias2"]:
        d[f"gru_{n}"], d[f"gru_d{n}"] = getattr(cell, n).data.copy(), getattr(cell, n).grad.copy()
    d["gru_x"], d["gru_h"], d["gru_out"], d["gru_dx"], d["gru_dh"] = gx, gh, out.data, a.grad, h.grad
    np.random.seed(4)
    fresh()
    gru = nn.GRU(3, 8, dtype=np.float32)
    sx = rng.standard_normal((5, 4, 3)).astype(np.float32)
    a = T(sx, True)
    out, hn = gru(a)
    (out * out).sum().backward()
    c = gru.GRUCells[0]
    for n in ["Wx1", "Wx2", "Wh1", "Wh2", "bias1", "bias2"]:
        d[f"gruseq_{n}"], d[f"gruseq_d{n}"] = getattr(c, n).data.copy(), getattr(c, n).grad.copy()
    d["gruseq_x"], d["gruseq_out"], d["gruseq_hn"], d["gruseq_dx"] = sx, out.data, hn.data, a.grad
    np.random.seed(5)
    fresh()
    rc = nn.RNNCell(6, 8, dtype=np.float32)
    a, h = T(gx, True), T(gh, True)
    out = rc(a, h)
    (out * out).sum().backward()
    for n in ["Wx", "Wh", "bias"]:
        d[f"rnn_{n}"], d[f"rnn_d{n}"] = getattr(rc, n).data.copy(), getattr(rc, n).grad.copy()
    d["rnn_out"], d["rnn_dx"], d["rnn_dh"] = out.data, a.grad, h.grad
    np.savez_compressed(os.path.join(OUT, "functional.npz"), **d)


def gen_adam():
    d = {}
    rng = np.random.default_rng(21)
    p1, p2 = rng.standard_normal((4, 3)).astype(np.float32), rng.standard_normal(5).astype(np.float32)
    fresh()
    a = nn.Parameter(pdn.Tensor(p1.copy(), dtype=np.float32))
    b = nn.Parameter(pdn.Tensor(p2.copy(), dtype=np.float32))
    opt = Adam([a, b], lr=1e-2, weight_decay=0.01)
    d["p1"], d["p2"] = p1, p2
    for t in range(3):
        g1, g2 = rng.standard_normal((4, 3)).astype(np.float32), rng.standard_normal(5).astype(np.float32)
        a.grad[...] = g1; b.grad[...] = g2
        opt.step()
        d[f"g1_{t}"], d[f"g2_{t}"], d[f"p1_{t}"], d[f"p2_{t}"] = g1, g2, a.data.copy(), b.data.copy()
    np.savez_compressed(os.path.join(OUT, "adam.npz"), **d)


def llama_run(V, D, H, Ff, L, B, layers, seed, steps, lr, full_tensors):
    from llm.llama.model import Llama
    fresh()
    np.random.seed(seed)
    m = Llama(V, D, H, Ff, 64 if full_tensors else 1024, B, layers, np.float32)
    m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(V, D)).astype(np.float32)
    ids = np.random.randint(0, V, (B, L))
    tgt = np.random.randint(0, V, (B, L))
    names = [n for n, p in m._parameters.items() if p.requires_grad]
    init = {n: m._parameters[n].data.copy() for n in names}
    opt = Adam(m.parameters(), lr=lr)
    losses, grads1 = [], None
    for s in range(steps):
        losses.append(m.finetune_step(ids, tgt, opt))
        if s == 0:
            grads1 = {n: m._parameters[n].grad.copy() for n in names}
    final = {n: m._parameters[n].data.copy() for n in names}
    return names, init, ids, tgt, losses, grads1, final


def gen_tiny_llama():
    names, init, ids, tgt, losses, g1, final = llama_run(64, 48, 2, 96, 16, 2, 2, 1234, 5, 1e-3, True)
    d = {"ids": ids, "tgt": tgt, "losses": np.array(losses, np.float64)}
    for n in names:
        d["init/" + n], d["grad1/" + n], d["final/" + n] = init[n], g1[n], final[n]
    np.savez_compressed(os.path.join(OUT, "tiny_llama.npz"), **d)
    print("tiny llama losses", losses)


def gen_full_llama():
    names, init, ids, tgt, losses, g1, final = llama_run(32000, 288, 6, 768, 256, 1, 6, 0, 3, 1e-4, False)
    out = {"config": dict(V=32000, D=288, H=6, F=768, L=256, B=1, layers=6, seed=0, lr=1e-4),
           "losses": [float(x) for x in losses],
           "grad1_norm": {n: float(np.linalg.norm(g1[n].astype(np.float64))) for n in names},
           "grad1_sum": {n: float(g1[n].astype(np.float64).sum()) for n in names},
           "final_norm": {n: float(np.linalg.norm(final[n].astype(np.float64))) for n in names}}
    json.dump(out, open(os.path.join(OUT, "llama_full.json"), "w"), indent=1)
    print("full llama losses", losses)


def gen_mlp_lenet():
    sys.argv = ["x"]
    d = {}
    # models restated from examples/pydynet/mnist.py:65-98 using the REFERENCE's nn layers
    class MLP(nn.Module):
        def __init__(s):
            super().__init__()
            s.layer1 = nn.Linear(784, 1024, dtype=np.float32)
            s.layer2 = nn.Linear(1024, 1024, dtype=np.float32)
            s.layer3 = nn.Linear(1024, 10, dtype=np.float32)

        def forward(s, x):
            x = x.reshape(x.shape[0], -1)
            return s.layer3(F.relu(s.layer2(F.relu(s.layer1(x)))))

    class LeNet(nn.Module):
        def __init__(s):
            super().__init__()
            s.conv1 = nn.Conv2d(3, 20, 3, 1, 1, dtype=np.float32)
            s.conv2 = nn.Conv2d(20, 50, 3, 1, 1, dtype=np.float32)
            s.fc1 = nn.Linear(8 * 8 * 50, 500, dtype=np.float32)
            s.fc2 = nn.Linear(500, 10, dtype=np.float32)

        def forward(s, x):
            x = F.max_pool2d(F.relu(s.conv1(x)), 2, 2)
            x = F.max_pool2d(F.relu(s.conv2(x)), 2, 2)
            x = x.reshape(-1, 8 * 8 * 50)
            return s.fc2(F.relu(s.fc1(x)))

    for name, cls, shape, B in [("mlp", MLP, (1, 28, 28), 32), ("lenet", LeNet, (3, 32, 32), 8)]:
        fresh()
        np.random.seed(42)
        net = cls()
        X = np.random.rand(B, *shape).astype(np.float32)
        y = np.random.randint(0, 10, B)
        opt = Adam(net.parameters(), lr=1e-4)
        losses, gn = [], None
        for s in range(3):
            loss = F.cross_entropy_loss(net(pdn.Tensor(X, dtype=np.float32)), pdn.Tensor(y, dtype=np.int64))
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.item())
            if s == 0:
                gn = {n: p.grad.copy() for n, p in net._parameters.items()}
        d[f"{name}_X"], d[f"{name}_y"], d[f"{name}_losses"] = X, y, np.array(losses)
        for n, g in gn.items():
            d[f"{name}_gnorm/{n}"] = np.float64(np.linalg.norm(g.astype(np.float64)))
            if g.size <= 20000:
                d[f"{name}_grad1/{n}"] = g
        for n, p in net._parameters.items():
            d[f"{name}_pnorm3/{n}"] = np.float64(np.linalg.norm(p.data.astype(np.float64)))
        print(name, losses)
    np.savez_compressed(os.path.join(OUT, "mlp_lenet.npz"), **d)


def gen_transformer():
    """The CoLA-shaped 1-layer Transformer (examples/pydynet/transformer.py) on the real reference:
    pins leading-axis LayerNorm with running statistics, the in-place padding mask and
    Embedding(padding_idx)."""
    sys.path.insert(0, os.path.dirname(OUT))
    import models_transformer as mt
    fresh()
    Transformer, loss_fn = mt.build(pdn, nn, F)
    c = mt.CFG
    ids, labels, emb = mt.make_inputs()
    np.random.seed(11)
    net = Transformer(c["embed"], c["layers"], c["heads"], c["expansion"], c["vocab"], c["max_len"])
    net.word_embedding.weight.data[...] = emb
    opt = Adam(net.parameters(), lr=c["lr"])
    d, losses = {}, []
    net.train()
    for s in range(c["steps"]):
        loss = loss_fn(net, pdn.Tensor(ids), pdn.Tensor(labels))
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.item())
        if s == 0:
            for n, p in net._parameters.items():
                if p.requires_grad:
                    d[f"gnorm/{n}"] = np.float64(np.linalg.norm(p.grad.astype(np.float64)))
                    if p.grad.size <= 5000:
                        d[f"grad1/{n}"] = p.grad.copy()
    d["losses"] = np.array(losses)
    for n, p in net._parameters.items():
        if "running" in n:
            d[f"final/{n}"] = p.data.copy()
        d[f"pnorm/{n}"] = np.float64(np.linalg.norm(p.data.astype(np.float64)))
    net.eval()
    with pdn.no_grad():
        mask = pdn.unsqueeze(pdn.Tensor(ids).eq(0), (1, 2)).astype(np.float32)
        d["eval_out"] = net(pdn.Tensor(ids), mask).data.copy()        # eval mode: running statistics
    pdn.autograd.set_grad_enabled(True)
    print("transformer", losses)
    np.savez_compressed(os.path.join(OUT, "transformer_example.npz"), **d)


def gen_dropout_bn():
    """examples/pydynet/dropout_bn.py: plain / Dropout / BatchNorm1d classifiers trained jointly."""
    sys.path.insert(0, os.path.dirname(OUT))
    import models_dropout_bn as md
    fresh()
    np.random.seed(42)
    d = md.run(pdn, nn, F, Adam)
    print("dropout_bn", d["losses"][-1])
    np.savez_compressed(os.path.join(OUT, "dropout_bn.npz"), **d)


def gen_ts_prediction():
    """examples/pydynet/ts_prediction.py (GRU regressor) and autograd1d.py on the real reference."""
    sys.path.insert(0, os.path.dirname(OUT))
    import models_ts_prediction as mt
    fresh()
    np.random.seed(7)
    d = mt.run(pdn, nn, Adam)
    fresh()
    d["autograd1d"] = mt.autograd1d(pdn)
    print("ts_prediction", d["losses"], "autograd1d ->", d["autograd1d"][-1])
    np.savez_compressed(os.path.join(OUT, "ts_prediction.npz"), **d)


def gen_misc():
    """pool1d, LSTM, RNN, SGD / Adagrad / Adadelta, LR schedulers on the real reference."""
    sys.path.insert(0, os.path.dirname(OUT))
    import models_misc as mm
    import pydynet.optim as optim
    from pydynet.optim import lr_scheduler
    fresh()
    d = mm.run(pdn, nn, F, optim, lr_scheduler)
    print("misc", len(d), "arrays; lr/cos", d["lr/cos"][:4])
    np.savez_compressed(os.path.join(OUT, "misc_layers.npz"), **d)


def gen_autograd2d():
    """examples/pydynet/autograd2d.py:5-33 (config 1): 30 GD steps on 0.5 x^T A x + b^T x."""
    fresh()
    A = pdn.Tensor([[3, 1.], [1, 2.]]); b = pdn.Tensor([-1., 1])
    np.random.seed(42)
    x = pdn.randn(2, requires_grad=True)
    traj = []
    for _ in range(30):
        obj = x @ A @ x / 2 + b @ x
        traj.append([*x.data.tolist(), float(obj.item())])
        obj.backward()
        x.data -= 0.1 * x.grad
        x.zero_grad()
    json.dump({"trajectory": traj}, open(os.path.join(OUT, "autograd2d.json"), "w"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_ops(); gen_functional(); gen_adam(); gen_tiny_llama(); gen_mlp_lenet(); gen_autograd2d(); gen_transformer(); gen_dropout_bn(); gen_ts_prediction(); gen_misc()
    gen_full_llama()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
