import sys, os
sys.path.insert(0, os.getcwd())
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
def timed(fn, it=20):
    fn(); hp.synchronize()
    with hp.Timer() as t:
        for _ in range(it): fn()
    return t.ms / it * 1e3
for rep in range(2):
    for name, c, s in (("rope", C._ptr, S._ptr), ("no rope", None, None)):
        f = timed(lambda: L_.call("pdn_attention_fwd_f32", q, k, v, o._ptr, lse._ptr, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, 1, c, s, st))
        b = timed(lambda: L_.call("pdn_attention_bwd_f32", q, k, v, o._ptr, do._ptr, lse._ptr, dq, dk, dv, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, 1, c, s, ws, wsb, st))
        print(f"{name:8s} fwd {f:7.1f} us  bwd {b:7.1f} us")
