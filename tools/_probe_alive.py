import sys, gc
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from pydynet_amd import hipnp as hp, _lib
import pydynet_amd as pdn
from pydynet_amd.llm.llama import Llama
from pydynet_amd.optim import Adam
lib = _lib.lib(); hp.set_device(0)
def run(tag):
    r = bench.llama_other_width(512, 8, 1536, steps=5, warmup=3)
    print(tag, round(r["ms_per_step"], 2), flush=True)
run("fresh")
np.random.seed(0)
m = Llama(32000, 288, 6, 768, 1024, 1, 6, np.float32); m.to("hip:0")
run("288 model alive (no optimizer, no steps)")
opt = Adam(m.parameters(), lr=1e-4); opt.flatten_grads()
run("+ optimizer with flat grads")
rng = np.random.default_rng(1)
ids = pdn.Tensor(rng.integers(0, 32000, (512, 256)), dtype=np.int64, device="hip:0"); tgt = pdn.Tensor(rng.integers(0, 32000, (512*256,)), dtype=np.int64, device="hip:0")
m.train(True)
for _ in range(4):
    opt.zero_grad(); l = m.loss(ids, tgt); l.backward(); opt.step()
hp.synchronize()
run("+ 4 steps done, everything alive")
l = None; gc.collect()
run("loss dropped")
opt = None; gc.collect()
run("optimizer dropped")
m = None; gc.collect()
run("model dropped")
