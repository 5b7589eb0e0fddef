import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import pydynet_amd as pdn, pydynet_amd.nn as nn, pydynet_amd.nn.functional as F
from pydynet_amd import hipnp as hp
from pydynet_amd.optim import Adam
hp.set_device(0); np.random.seed(0)
T_, B_, Hd = 40, 1568, 32
gru = nn.GRU(1, Hd, dtype=np.float32).to("hip:0"); head = nn.Linear(Hd, 1, dtype=np.float32).to("hip:0")
opt = Adam(list(gru.parameters()) + list(head.parameters()), lr=1e-3)
xs = pdn.Tensor(np.random.rand(T_, B_, 1).astype(np.float32), device="hip:0")
ys = pdn.Tensor(np.random.rand(B_, 1).astype(np.float32), device="hip:0")
for _ in range(8):
    out, hn = gru(xs); loss = F.mse_loss(head(hn[0]), ys)
    opt.zero_grad(); loss.backward(); opt.step()
hp.synchronize(); print(loss.item())
