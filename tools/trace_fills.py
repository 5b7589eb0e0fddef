"""Which Python call sites issue the hipMemset / device-to-device copy launches of a training step (pdn_fill with 0 /
pdn_cast of a contiguous same-dtype array): one benchmark-shaped step, call stacks counted.  usage: python tools/trace_fills.py [batch=64]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydynet_amd as pdn
from pydynet_amd import _lib, hipnp
from pydynet_amd.llm.llama import Llama
from pydynet_amd.optim import Adam

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hipnp.set_device(0)
L = _lib.lib()
np.random.seed(0)
m = Llama(32000, 288, 6, 768, 1024, 1, 6, np.float32)
m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(32000, 288)).astype(np.float32)
m.to("hip:0")
opt = Adam(m.parameters(), lr=1e-4)
opt.flatten_grads()
rng = np.random.default_rng(0)
ids = pdn.Tensor(rng.integers(0, 32000, (B, 256)), dtype=np.int64, device="hip:0")
tgt = pdn.Tensor(rng.integers(0, 32000, (B * 256,)), dtype=np.int64, device="hip:0")


def step():
    opt.zero_grad(); loss = m.loss(ids, tgt); loss.backward(); opt.step()


step(); step()
sites = collections.Counter()
orig = L.call


def spy(name, *a):
    if name in ("pdn_fill", "pdn_cast", "pdn_memset", "pdn_memcpy_d2d"):
        st = traceback.extract_stack()[:-1]
        key = " <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in st[-6:])
        sites[(name, key)] += 1
    return orig(name, *a)


L.call = spy
step()
L.call = orig
hipnp.synchronize()
for (n, k), c in sites.most_common(40):
    print(c, n, k)
