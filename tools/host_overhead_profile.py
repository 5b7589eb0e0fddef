"""Where the HOST time of an eager step goes (cProfile over the Transformer example's step at its own shape: ~100 launches,
~1.5 ms of kernels, ~1.9 ms of Python + ctypes when issued eagerly).  usage: python tools/host_overhead_profile.py [steps=200]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydynet_amd as pdn
import pydynet_amd.nn as nn
import pydynet_amd.nn.functional as F
from pydynet_amd import hipnp as hp
from pydynet_amd.optim import Adam
from pydynet_amd.core.tensor import Graph
from tests import models_transformer as mt

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hp.set_device(0)
Transformer, loss_fn = mt.build(pdn, nn, F)
B, L, V, D, H, E = 128, 44, 6000, 512, 4, 3
rng = np.random.default_rng(0)
ids_np = rng.integers(1, V, (B, L))
labels_np = rng.choice([-1.0, 1.0], B).astype(np.float32)
Graph.clear()
np.random.seed(0)
net = Transformer(D, 1, H, E, V, L)
net.to("hip:0")
opt = Adam(net.parameters(), lr=5e-4)
opt.flatten_grads()
net.train()
ids, labels = pdn.Tensor(ids_np, dtype=np.int64, device="hip:0"), pdn.Tensor(labels_np, device="hip:0")


def step():
    loss = loss_fn(net, ids, labels)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss


for _ in range(20):
    step()
hp.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
hp.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
