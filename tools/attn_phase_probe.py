"""Causal vs non-causal timing of the resident attention kernels on the benchmark shape: the two differ by 28 of 64
tile pairs per head and by nothing else, which separates the per-pair loop time from the per-workgroup fixed cost
(staging K,V / Q,dO through LDS, operand loads, stores)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib
hp.set_device(0)
L_ = _lib.lib()
B, H, L, hd = 256, 6, 256, 48
D = H * hd
rng = np.random.default_rng(0)
qkv = hp.from_numpy(rng.standard_normal((B * L, 3 * D), dtype=np.float32))
do = hp.from_numpy(rng.standard_normal((B, L, H, hd), dtype=np.float32))
inv = 1.0 / (10000 ** (np.arange(0, hd, 2) / hd))
fr = np.outer(np.arange(L), inv).astype(np.float32)
C, S = hp.from_numpy(np.cos(fr)), hp.from_numpy(np.sin(fr))
o, lse = hp.empty((B, L, H, hd)), hp.empty((B, H, L))
dqkv = hp.empty((B * L, 3 * D))
q, k, v = qkv._ptr, qkv._ptr + 4 * D, qkv._ptr + 8 * D
dq, dk, dv = dqkv._ptr, dqkv._ptr + 4 * D, dqkv._ptr + 8 * D
ws, wsb = hp.workspace(4 * B * H * L)
st = hp.stream()


def timed(fn, it=10):
    fn(); hp.synchronize()
    with hp.Timer() as t:
        for _ in range(it):
            fn()
    return t.ms / it * 1e3


res = {}
for causal in (1, 0):
    f = timed(lambda: L_.call("pdn_attention_fwd_f32", q, k, v, o._ptr, lse._ptr, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, causal, C._ptr, S._ptr, st))
    b = timed(lambda: L_.call("pdn_attention_bwd_f32", q, k, v, o._ptr, do._ptr, lse._ptr, dq, dk, dv, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, causal, C._ptr, S._ptr, ws, wsb, st))
    res[causal] = (f, b)
    print(f"causal={causal}: fwd {f:7.1f} us   bwd {b:7.1f} us")
for name, i, mf in (("fwd", 0, 56), ("bwd (dq + dkv)", 1, 192)):
    per_pair = (res[0][i] - res[1][i]) / 28.0
    fixed = res[1][i] - 36 * per_pair
    # 1536 heads on 256 CUs: 6 heads per CU; a pair's MFMAs at full rate: mf * 64 cycles / 4 SIMDs per head
    ideal = 6 * mf * 64 / 4 / 2.4e3
    print(f"{name}: {per_pair:6.2f} us per tile pair (matrix pipe alone: {ideal:5.2f}), fixed {fixed:6.1f} us of {res[1][i]:6.1f}")
