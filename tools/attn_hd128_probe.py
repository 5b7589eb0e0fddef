"""Head dim 128 attention (csrc/attention_hd128.hip) timed alone through the C entry points: forward and backward (dQ + dK/dV
kernels), against the streaming kernels of the same shape.  "useful" = the FLOPs of the unmasked part only (forward 4 L^2 hd
per head, half of it under a causal mask; backward 2.5 x) over the fp32-MFMA peak.
usage: python tools/attn_hd128_probe.py [L=256] [heads=2048] [causal=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib

hp.set_device(0)
Lb = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BH = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
causal = int(sys.argv[3]) if len(sys.argv) > 3 else 1
H, HD = 4, 128
B = BH // H
rng = np.random.default_rng(0)
q, k, v, go = (hp.from_numpy(rng.standard_normal((B, L, H, HD), dtype=np.float32)) for _ in range(4))
o, dq, dk, dv = (hp.empty((B, L, H, HD), np.float32) for _ in range(4))
lse = hp.empty((B, H, L), np.float32)
rs, bs = H * HD, L * H * HD
ws, wsb = hp.workspace(Lb.query("pdn_attention_bwd_workspace_bytes", B, H, L))
st = hp.stream()


def t(fn, it=60):
    for _ in range(20):
        fn()
    hp.synchronize()
    with hp.Timer() as tm:
        for _ in range(it):
            fn()
    return tm.ms / it * 1e3


useful = 4.0 * L * L * HD * BH * (0.5 if causal else 1.0)
fr = lambda us, fl: 100 * fl / us / 1e-6 / 157.3e12
f_res = lambda: Lb.call("pdn_attention_fwd_f32", q._ptr, k._ptr, v._ptr, o._ptr, lse._ptr, B, H, L, HD, rs, bs, rs, bs, causal, None, None, st)
b_res = lambda: Lb.call("pdn_attention_bwd_f32", q._ptr, k._ptr, v._ptr, o._ptr, go._ptr, lse._ptr, dq._ptr, dk._ptr, dv._ptr, B, H, L,
                        HD, rs, bs, rs, bs, causal, None, None, ws, wsb, st)
f_str = lambda: Lb.call("pdn_attention_stream_fwd_f32", q._ptr, k._ptr, v._ptr, o._ptr, lse._ptr, B, H, L, L, HD, rs, bs, rs, bs, causal, 0,
                        None, 0, 0, 0, 0, None, None, st)
tf, tb, ts = t(f_res), t(b_res), t(f_str)
print(f"hd 128, L {L}, {BH} heads, causal {causal}: resident forward {tf:8.1f} us = {fr(tf, useful):5.1f} % useful, backward {tb:8.1f} us = "
      f"{fr(tb, 2.5 * useful):5.1f} %; streaming forward {ts:8.1f} us = {fr(ts, useful):5.1f} %", flush=True)
