"""Launch-bound regimes with and without hipGraph replay (hipnp.Graph): MLP B=256, LeNet B=256,
GRU(1->32, T=40, B=1568), Llama at small per-GPU batches.  usage: python tools/bench_graph.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydynet_amd as pdn
import pydynet_amd.nn as nn
import pydynet_amd.nn.functional as F
from pydynet_amd import hipnp as hp
from pydynet_amd.optim import Adam
from pydynet_amd.core.tensor import Graph
from pydynet_amd.llm.llama import Llama

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
hp.set_device(0)


def timeit(name, step, unit, per_step):
    for _ in range(3):
        step()
    hp.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    hp.synchronize()
    eager = (time.perf_counter() - t0) / steps
    g = hp.Graph()
    g.capture(step)
    g.replay(); hp.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    hp.synchronize()
    graph = (time.perf_counter() - t0) / steps
    print(f"{name:34s} eager {eager * 1e3:8.3f} ms/step   graph {graph * 1e3:8.3f} ms/step ({g.nodes} nodes)   "
          f"{per_step / graph:12.0f} {unit}/s", flush=True)
    g.destroy()


class MLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.layer1 = nn.Linear(784, 1024, dtype=np.float32)
        self.layer2 = nn.Linear(1024, 1024, dtype=np.float32)
        self.layer3 = nn.Linear(1024, 10, dtype=np.float32)

    def forward(self, x):
        x = x.reshape(x.shape[0], -1)
        return self.layer3(F.relu(self.layer2(F.relu(self.layer1(x)))))


class LeNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 20, 3, 1, 1, dtype=np.float32)
        self.conv2 = nn.Conv2d(20, 50, 3, 1, 1, dtype=np.float32)
        self.fc1 = nn.Linear(8 * 8 * 50, 500, dtype=np.float32)
        self.fc2 = nn.Linear(500, 10, dtype=np.float32)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.conv1(x)), 2, 2)
        x = F.max_pool2d(F.relu(self.conv2(x)), 2, 2)
        return self.fc2(F.relu(self.fc1(x.reshape(-1, 8 * 8 * 50))))


for name, cls, shape, B in (("mlp B=256", MLP, (1, 28, 28), 256), ("lenet B=256", LeNet, (3, 32, 32), 256)):
    Graph.clear(); np.random.seed(42)
    net = cls().to("hip:0")
    opt = Adam(net.parameters(), lr=1e-4)
    X = pdn.Tensor(np.random.rand(B, *shape).astype(np.float32), device="hip:0")
    y = pdn.Tensor(np.random.randint(0, 10, B), dtype=np.int64, device="hip:0")

    def step():
        loss = F.cross_entropy_loss(net(X), y)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
    timeit(name, step, "samples", B)

Graph.clear(); np.random.seed(0)
T_, B_, Hd = 40, 1568, 32
gru = nn.GRU(1, Hd, dtype=np.float32).to("hip:0")
head = nn.Linear(Hd, 1, dtype=np.float32).to("hip:0")
opt = Adam(list(gru.parameters()) + list(head.parameters()), lr=1e-3)
xs = pdn.Tensor(np.random.rand(T_, B_, 1).astype(np.float32), device="hip:0")
ys = pdn.Tensor(np.random.rand(B_, 1).astype(np.float32), device="hip:0")


def gstep():
    out, hn = gru(xs)
    loss = F.mse_loss(head(hn[0]), ys)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss


timeit("gru T=40 B=1568 H=32", gstep, "sequences", B_)

V, D, H, F_, L, LAYERS = 32000, 288, 6, 768, 256, 6
for B in (8, 64):
    Graph.clear(); np.random.seed(0)
    model = Llama(V, D, H, F_, 1024, 1, LAYERS, np.float32)
    model.tok_embedding.weight.data[...] = (0.02 * np.random.randn(V, D)).astype(np.float32)
    model.to("hip:0")
    opt = Adam(model.parameters(), lr=1e-4)
    opt.flatten_grads()
    rng = np.random.default_rng(1000)
    ids = pdn.Tensor(rng.integers(0, V, (B, L)), dtype=np.int64, device="hip:0")
    tgt = pdn.Tensor(rng.integers(0, V, (B * L,)), dtype=np.int64, device="hip:0")
    model.train(True)

    def lstep():
        opt.zero_grad()
        loss = model.loss(ids, tgt)
        loss.backward()
        opt.step()
        return loss
    timeit(f"llama B={B}", lstep, "samples", B)
    del model, opt
