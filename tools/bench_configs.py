"""Throughput of BASELINE.json configs 2 (MNIST-shaped MLP) and 3 (LeNet on 3x32x32) on one GPU
(SURVEY 8d).  Prints one line per (config, batch): samples/s, achieved GEMM TFLOP/s from the
algorithmic FLOP count, and for LeNet the HBM GB/s of the algorithmic conv bytes.
usage: python tools/bench_configs.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as _hpsync
import pydynet_amd as pdn
import pydynet_amd.nn as nn
import pydynet_amd.nn.functional as F
from pydynet_amd.optim import Adam
from pydynet_amd.core.tensor import Graph

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = sys.argv[2] if len(sys.argv) > 2 else None          # e.g. "lenet:4096"


class MLP(nn.Module):                      # examples/pydynet/mnist.py:65-79
    def __init__(self):
        super().__init__()
        self.layer1 = nn.Linear(784, 1024, dtype=np.float32)
        self.layer2 = nn.Linear(1024, 1024, dtype=np.float32)
        self.layer3 = nn.Linear(1024, 10, dtype=np.float32)

    def forward(self, x):
        x = x.reshape(x.shape[0], -1)
        return self.layer3(F.relu(self.layer2(F.relu(self.layer1(x)))))


class LeNet(nn.Module):                    # mnist.py:82-98, shape-adapted to 3x32x32
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


CASES = [("mlp", MLP, (1, 28, 28), 9_564_160, (256, 8192, 65536)),
         ("lenet", LeNet, (3, 32, 32), 25_665_840, (256, 4096))]
for name, cls, shape, flops, batches in CASES:
    for B in batches:
        if only and only != f"{name}:{B}":
            continue
        Graph.clear()
        np.random.seed(42)
        net = cls().to("cuda")
        opt = Adam(net.parameters(), lr=1e-4)
        X = pdn.Tensor(np.random.rand(B, *shape).astype(np.float32), device="cuda")
        y = pdn.Tensor(np.random.randint(0, 10, B), dtype=np.int64, device="cuda")

        def step():
            loss = F.cross_entropy_loss(net(X), y)
            opt.zero_grad(); loss.backward(); opt.step()
            return loss

        for _ in range(3):
            step()
        _hpsync.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        _hpsync.synchronize()
        dt = (time.perf_counter() - t0) / steps
        line = f"{name:6s} B={B:6d}  {dt*1e3:8.3f} ms/step  {B/dt:12.0f} samples/s  {flops*B/dt/1e12:7.2f} TFLOP/s (algorithmic)  loss {loss.item():.4f}"
        if name == "lenet":
            conv_bytes = 3 * (12 + 80 + 20 + 50 + 12.5) * 1024     # fwd bytes/sample x ~3 for fwd+bwd
            line += f"  conv traffic {conv_bytes*B/dt/1e9:7.1f} GB/s (algorithmic)"
        print(line, flush=True)

# ---- GRU sequence of examples/pydynet/ts_prediction.py: GRU(1 -> 32), T = 40, batch ~1568 -------------
if not only or only.startswith("gru"):
    Graph.clear()
    np.random.seed(0)
    T_, B_, Hd = 40, 1568, 32
    gru = nn.GRU(1, Hd, dtype=np.float32).to("cuda")
    head = nn.Linear(Hd, 1, dtype=np.float32).to("cuda")
    opt = Adam(list(gru.parameters()) + list(head.parameters()), lr=1e-3)
    xs = pdn.Tensor(np.random.rand(T_, B_, 1).astype(np.float32), device="cuda")
    ys = pdn.Tensor(np.random.rand(B_, 1).astype(np.float32), device="cuda")

    def gstep():
        out, hn = gru(xs)
        loss = F.mse_loss(head(hn[0]), ys)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss

    for _ in range(3):
        gstep()
    _hpsync.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = gstep()
    _hpsync.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"gru    T={T_} B={B_} H={Hd}  {dt*1e3:8.3f} ms/step  {B_/dt:12.0f} sequences/s  loss {loss.item():.4f}", flush=True)
