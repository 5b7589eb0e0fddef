"""Why is a forced single-rank data-parallel step slower?  Times the bench step under variants:
  base        no communicator
  init        RCCL communicator created, no collectives, no hooks
  dp_nocomm   DataParallel wrapper (flat buckets, hooks) but collectives disabled
  dp          full forced-DP path
usage: python tools/dp_overhead_probe.py <variant> [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydynet_amd as pdn
from pydynet_amd import hipnp, distributed as pdist
from pydynet_amd.llm.llama import Llama
from pydynet_amd.optim import Adam
from pydynet_amd.distributed import DataParallel

variant = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
V, D, H, F_, L, LAYERS = 32000, 288, 6, 768, 256, 6
hipnp.set_device(0)
if variant != "base" and variant != "streams":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    pdist.init_process_group("rccl", 0)
np.random.seed(0)
model = Llama(V, D, H, F_, 1024, 1, LAYERS, np.float32)
model.tok_embedding.weight.data[...] = (0.02 * np.random.randn(V, D)).astype(np.float32)
model.to("hip:0")
opt = Adam(model.parameters(), lr=1e-4)
dp = None
if variant in ("dp", "dp_events", "dp_same"):
    g = pdist.get_group()
    if variant == "dp_events":            # the event dance without any RCCL call
        g._L_real_call = g._L.call
        class _NoColl:
            def __init__(self, L): self.L = L
            def call(self, name, *a):
                if name.startswith("pdn_comm_all") or name == "pdn_comm_broadcast":
                    return
                return self.L.call(name, *a)
            def __getattr__(self, k): return getattr(self.L, k)
        g._L = _NoColl(g._L)
    if variant == "dp_same":              # collectives on the compute stream, no events
        g._stream = hipnp.stream()
        g._after_compute = lambda: None
        g._mark = lambda: None
        g.wait = lambda: None
    dp = DataParallel(model, opt, always_reduce=True)
elif variant == "dp_nocomm":
    dp = DataParallel(model, opt, always_reduce=False)
else:
    opt.flatten_grads()
rng = np.random.default_rng(1000)
ids = pdn.Tensor(rng.integers(0, V, (B, L)), dtype=np.int64, device="hip:0")
tgt = pdn.Tensor(rng.integers(0, V, (B * L,)), dtype=np.int64, device="hip:0")
model.train(True)


def step():
    opt.zero_grad()
    loss = model.loss(ids, tgt)
    loss.backward()
    if dp is not None:
        dp.finish()
    opt.step()
    return loss


for _ in range(3):
    step()
hipnp.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    l = step()
hipnp.synchronize()
print(f"{variant:10s} B={B} {(time.perf_counter() - t0) / 8 * 1e3:8.2f} ms/step  loss {l.item():.4f}", flush=True)
