"""Run the fused attention forward/backward a few times (for rocprofv3 --pmc runs).  usage: attn_one.py [iters] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as _hpsync
from pydynet_amd import hipnp as hp, _lib
hp.set_device(0)
L_ = _lib.lib()
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H, L, hd = 6, 256, 48
rnd = lambda *s: hp.from_numpy(np.random.default_rng(0).standard_normal(s, dtype=np.float32))
q, k, v, do = rnd(B, L, H, hd), rnd(B, L, H, hd), rnd(B, L, H, hd), rnd(B, L, H, hd)
o, lse = hp.empty((B, L, H, hd)), hp.empty((B, H, L))
dq, dk, dv = hp.empty((B, L, H, hd)), hp.empty((B, L, H, hd)), hp.empty((B, L, H, hd))
ws, wsb = hp.workspace(L_.query("pdn_attention_bwd_workspace_bytes", B, H, L))
for _ in range(it):
    L_.call("pdn_attention_fwd_f32", q._ptr, k._ptr, v._ptr, o._ptr, lse._ptr, B, H, L, hd, H * hd, L * H * hd, H * hd, L * H * hd, 1,
            None, None, hp.stream())
    L_.call("pdn_attention_bwd_f32", q._ptr, k._ptr, v._ptr, o._ptr, do._ptr, lse._ptr, dq._ptr, dk._ptr, dv._ptr,
            B, H, L, hd, H * hd, L * H * hd, H * hd, L * H * hd, 1, None, None, ws, wsb, hp.stream())
_hpsync.synchronize()
print("done")
