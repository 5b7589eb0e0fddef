"""Resident (K/V of a head in LDS 256 rows at a time; hd 48 / 64, L <= 1024) vs streaming attention kernels:
B*H heads, causal, RoPE in the loads, q/k/v as column blocks of a packed projection.
usage: attn_compare.py [batch [L [head_dim]]]   (default 256 256 48 = the benchmark shape, 1536 heads);
also prints the time per causal tile pair, the measure the sequence lengths are compared by."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib
hp.set_device(0)
L_ = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, L, hd = 6, int(sys.argv[2]) if len(sys.argv) > 2 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 48
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
qd, dqd = hp.empty((B * L, D)), hp.empty((B * L, D))
ws, wsb = hp.workspace(4 * B * H * L)
st = hp.stream()
flops_fwd = 4.0 * L * L * hd * B * H / 2          # causal-useful


def timed(fn, it=10):
    fn(); hp.synchronize()
    with hp.Timer() as t:
        for _ in range(it):
            fn()
    return t.ms / it * 1e3


r_f = timed(lambda: L_.call("pdn_attention_fwd_f32", q, k, v, o._ptr, lse._ptr, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, 1, C._ptr, S._ptr, st))
r_b = timed(lambda: L_.call("pdn_attention_bwd_f32", q, k, v, o._ptr, do._ptr, lse._ptr, dq, dk, dv, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, 1, C._ptr, S._ptr, ws, wsb, st))
s_f = timed(lambda: L_.call("pdn_attention_stream_fwd_f32", qd._ptr, k, v, o._ptr, lse._ptr, B, H, L, L, hd, D, L * D, 3 * D, L * 3 * D, 1, 0, None, 0, 0, 0, 0, C._ptr, S._ptr, st))
s_b = timed(lambda: L_.call("pdn_attention_stream_bwd_f32", qd._ptr, k, v, o._ptr, do._ptr, lse._ptr, dqd._ptr, dk, dv, B, H, L, L, hd, D, L * D, 3 * D, L * 3 * D, 1, 0, None, 0, 0, 0, 0, C._ptr, S._ptr, ws, wsb, st))
rows = [("resident", r_f, r_b), ("stream", s_f, s_b)]
if hd == 48 and (L <= 256 or (L % 256 == 0 and L <= 1024)):
    # (round 5: 512 / 768 / 1024 positions run as 256-row block pairs on the same kernels, csrc/attention_blocks.hip)
    # round 4: q, k rotated by the projection's epilogue -> the persistent, DMA-staged kernels (csrc/attention_p.hip)
    p_f = timed(lambda: L_.call("pdn_attention_fwd_f32", q, k, v, o._ptr, lse._ptr, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, 1, None, None, st))
    p_b = timed(lambda: L_.call("pdn_attention_bwd_rotated_f32", q, k, v, o._ptr, do._ptr, lse._ptr, dq, dk, dv, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, 1, C._ptr, S._ptr, ws, wsb, st))
    rows.insert(0, ("persistent", p_f, p_b))
pairs = (L // 32) * (L // 32 + 1) / 2 * B * H
print(f"B*H = {B * H} heads, L = {L}, hd = {hd}: {pairs:.0f} causal tile pairs")
for name, f, b in rows:
    print(f"{name:10s} {1e3 * f / pairs:6.3f} / {1e3 * b / pairs:6.3f} ns per pair   fwd {f:8.1f} us ({flops_fwd / f / 1e6:6.1f} TFLOP/s causal-useful = {100 * flops_fwd / f / 1e6 / 157.3:4.1f} % of fp32 MFMA)   bwd {b:8.1f} us", flush=True)
