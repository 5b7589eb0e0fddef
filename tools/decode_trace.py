"""Reads the phase timestamps a -DDEC_TRACE build of the decode kernels leaves (run through tools/decode_trace.sh):
per kernel kind the mean duration of every phase of workgroup 0, and the gaps between consecutive kernels of a token
(end of one kernel's workgroup 0 -> start of the next one's).  100 MHz clock: 10 ns resolution."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib
import pydynet_amd as pdn
from pydynet_amd.llm.llama import Llama

KIND = {0: "gemv q|k|v", 1: "attention+oproj", 2: "pick", 3: "mlp", 4: "gemv lm_head", 5: "block (range 0)"}
PHASES = {0: ["issue prefetch", "stage+norm", "fma+shuffle", "reduce+store"],
          4: ["issue prefetch", "stage+norm", "fma+shuffle", "reduce+store"],
          1: ["pos + issue Wo", "rope/append", "scores+max", "exp+sum", "p.v", "combine", "oproj"],
          2: ["candidates", "emb row"],
          3: ["issue prefetch", "stage sum", "norm", "gate|up fma", "swiglu", "down+store"],
          5: ["issue", "stage+norm", "q|k|v", "rope", "scores+max", "exp, p.v, combine", "oproj"]}

np.random.seed(0)
model = Llama(32000, 288, 6, 768, 1024, 1, 6, np.float32)
model.tok_embedding.weight.data[...] = (0.02 * np.random.randn(32000, 288)).astype(np.float32)
model = model.to("cuda")
model.eval()
ids = np.random.randint(0, 32000, (1, 8))
cdll = _lib.lib().cdll
SLOTS = 8192


def dump(name):
    buf = np.zeros((SLOTS, 10), np.uint64)
    n = ctypes.c_uint(0)
    getattr(cdll, name)(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    return buf[:min(n.value, SLOTS)]


with pdn.no_grad():
    for _ in model.generate(ids, 60):          # warm-up, capture
        pass
    hp.synchronize()
    dump("pdn_dec_trace_dump_step"); dump("pdn_dec_trace_dump_layer"); dump("pdn_dec_trace_dump_block")
    for _ in model.generate(ids, 200):
        pass
    hp.synchronize()
rows = np.concatenate([dump("pdn_dec_trace_dump_step"), dump("pdn_dec_trace_dump_layer"), dump("pdn_dec_trace_dump_block")])
rows = rows[np.argsort(rows[:, 0])]
rows = rows[len(rows) // 4:]                   # the prompt pass / first tokens are not graph replays
kinds = rows[:, 9].astype(int)
print(f"{len(rows)} traced kernels")
for k, name in KIND.items():
    r = rows[kinds == k]
    if not len(r):
        continue
    np_ = len(PHASES[k])
    t = r[:, :np_ + 1].astype(np.int64)
    d = np.diff(t, axis=1) * 0.01
    print(f"{name:16s} n={len(r):5d}  workgroup-0 span {d.sum(1).mean():5.2f} us: " +
          ", ".join(f"{p} {v:.2f}" for p, v in zip(PHASES[k], d.mean(0))))
# gaps: end stamp of kernel i (its last phase) -> first stamp of kernel i + 1
ends = np.array([rows[i, len(PHASES[kinds[i]])] for i in range(len(rows))], np.int64)
gap = (rows[1:, 0].astype(np.int64) - ends[:-1]) * 0.01
for k, name in KIND.items():
    sel = kinds[1:] == k
    if sel.any():
        print(f"gap before {name:16s}: {np.median(gap[sel]):5.2f} us (median)")
tok = rows[kinds == 2][:, 0].astype(np.int64)
print(f"token period {np.median(np.diff(tok)) * 0.01:.1f} us")
