"""Training-step throughput of a Llama of ANOTHER width than the benchmarked 288 (the reference's constructor is
general: Llama(vocab, embed_dim, n_heads, ffn_dim, ...), llm/llama/model.py:153-181), as % of the fp32-MFMA peak at
model level, with the kernel families its GEMMs took.
usage: bench_llama_dims.py [dim heads ffn [batch [seq [layers [vocab]]]]]     (default 512 8 1536 256 256 6 32000)"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib
import pydynet_amd as pdn
from pydynet_amd.llm.llama import Llama
from pydynet_amd.optim import Adam
a = [int(v) for v in sys.argv[1:]]
D, H, F, B, L, NL, V = (a + [512, 8, 1536, 256, 256, 6, 32000][len(a):])[:7]
lib = _lib.lib()
hp.set_device(0)
np.random.seed(0)
m = Llama(V, D, H, F, max(L, 256), 1, NL, np.float32)
m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(V, D)).astype(np.float32)
m.to("hip:0")
opt = Adam(m.parameters(), lr=1e-4)
opt.flatten_grads()
rng = np.random.default_rng(1)
ids = pdn.Tensor(rng.integers(0, V, (B, L)), dtype=np.int64, device="hip:0")
tgt = pdn.Tensor(rng.integers(0, V, (B * L,)), dtype=np.int64, device="hip:0")
m.train(True)


def step():
    opt.zero_grad(); loss = m.loss(ids, tgt); loss.backward(); opt.step(); return loss


for _ in range(3):
    step()
hp.synchronize()
lib.call("pdn_gemm_prof_enable", 1)
n = 6
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
hp.synchronize()
dt = (time.perf_counter() - t0) / n
ms, fl, cnt = (ctypes.c_double * 5)(), (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
lib.call("pdn_gemm_prof_enable", 0)
lib.call("pdn_gemm_prof_collect_families", ms, fl, cnt)
flop = 3 * L * (NL * (4 * 2 * D * D + 3 * 2 * D * F + 2 * 2 * L * D) + 2 * D * V)       # SURVEY 8d's count, general
print(f"Llama dim {D} heads {H} (hd {D // H}) ffn {F} layers {NL} vocab {V}, batch {B} x seq {L}: {1e3 * dt:.2f} ms/step, "
      f"{B / dt:.0f} samples/s, {100 * flop * B / dt / 157.3e12:.1f} % of the fp32-MFMA peak at model level, loss {loss.item():.4f}")
names = ("tiled", "tn_stream", "rowres", "outres", "outres_tn")
print("   GEMM families: " + ", ".join(f"{nm} {100 * fl[i] / (ms[i] * 1e-3) / 157.3e12 if ms[i] else 0:.1f} % ({100 * ms[i] * 1e-3 / (n * dt):.0f} % of the step)"
                                       for i, nm in enumerate(names) if cnt[i]))
