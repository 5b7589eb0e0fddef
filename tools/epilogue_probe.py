"""Fused epilogues of the row-resident projections (csrc/gemm_rowres.hip, round 4) against the launches they replace:
gate | up GEMM + swiglu_rows_fwd, dh GEMM + swiglu_rows_bwd, q | k | v GEMM (RoPE then costs the attention kernels).
usage: python tools/epilogue_probe.py [tokens=65536]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib

hp.set_device(0)
L = _lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K, F, D, Lq, hd = 288, 768, 288, 256, 48
rng = np.random.default_rng(0)
PEAK = 157.3e12


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters):
            fn()
    return t.ms / iters * 1e3


def stack(mats):
    buf = hp.empty((len(mats),) + mats[0].shape, np.float32)
    for i, m in enumerate(mats):
        buf[i] = hp.from_numpy(m)
    return buf


x = hp.from_numpy(rng.standard_normal((T, K), dtype=np.float32))
wgu = stack([(0.05 * rng.standard_normal((K, F))).astype(np.float32) for _ in range(2)])
wd = hp.from_numpy((0.05 * rng.standard_normal((F, K))).astype(np.float32))
dy = hp.from_numpy(rng.standard_normal((T, K), dtype=np.float32))
gu0, gu1, h0, h1 = hp.empty((T, 2 * F)), hp.empty((T, 2 * F)), hp.empty((T, F)), hp.empty((T, F))
halves = hp.ndarray(gu0._buf, gu0._ptr, (2, T, F), (F, 2 * F, 1), gu0.dtype)


def sep_fwd():
    hp.gemm(x, wgu, halves)
    L.call("pdn_swiglu_rows_fwd_f32", gu0._ptr, h0._ptr, T, F, hp.stream())


def fus_fwd():
    L.call("pdn_gateup_swiglu_fwd_f32", x._ptr, wgu._ptr, K * F, gu1._ptr, h1._ptr, T, F, K, K, hp.stream())


sep_fwd(); fus_fwd(); hp.synchronize()
e1 = np.abs(gu0.get()[:2048] - gu1.get()[:2048]).max(), np.abs(h0.get()[-2048:] - h1.get()[-2048:]).max() / np.abs(h0.get()[-2048:]).max()
t_g = bench(lambda: hp.gemm(x, wgu, halves))
t0, t1 = bench(sep_fwd), bench(fus_fwd)
fl = 2.0 * T * 2 * F * K
print(f"gate|up + SwiGLU fwd, {T} tokens: GEMM alone {t_g:7.1f} us, GEMM + swiglu_rows_fwd {t0:7.1f} us, fused {t1:7.1f} us "
      f"({100 * fl / t1 / 1e-6 / PEAK:4.1f} % of peak)   max |d gu| {e1[0]:.1e}  rel d h {e1[1]:.1e}", flush=True)

dh, dgu0, dgu1 = hp.empty((T, F)), hp.empty((T, 2 * F)), hp.empty((T, 2 * F))


def sep_bwd():
    hp.gemm(dy, wd.T, dh)
    L.call("pdn_swiglu_rows_bwd_f32", gu0._ptr, dh._ptr, dgu0._ptr, T, F, hp.stream())


def fus_bwd():
    L.call("pdn_swiglu_bwd_gemm_f32", dy._ptr, wd._ptr, gu0._ptr, dgu1._ptr, T, F, K, K, hp.stream())


sep_bwd(); fus_bwd(); hp.synchronize()
a, b = dgu0.get()[:2048], dgu1.get()[:2048]
e2 = np.abs(a - b).max() / np.abs(a).max()
t_g = bench(lambda: hp.gemm(dy, wd.T, dh))
t0, t1 = bench(sep_bwd), bench(fus_bwd)
fl = 2.0 * T * F * K
print(f"dh + SwiGLU bwd,      {T} tokens: GEMM alone {t_g:7.1f} us, GEMM + swiglu_rows_bwd {t0:7.1f} us, fused {t1:7.1f} us "
      f"({100 * fl / t1 / 1e-6 / PEAK:4.1f} % of peak)   rel d dgu {e2:.1e}", flush=True)

wqkv = stack([(0.05 * rng.standard_normal((K, D))).astype(np.float32) for _ in range(3)])
inv = 1.0 / (10000 ** (np.arange(0, hd, 2)[: hd // 2] / hd))
fr = np.outer(np.arange(Lq), inv)
cos, sin = hp.from_numpy(np.cos(fr).astype(np.float32)), hp.from_numpy(np.sin(fr).astype(np.float32))
tab = hp.empty((Lq, hd, 2), np.float32)
L.call("pdn_rope_table_f32", cos._ptr, sin._ptr, tab._ptr, Lq, hd, hp.stream())
q0, q1 = hp.empty((T, 3 * D)), hp.empty((T, 3 * D))
blocks = hp.ndarray(q0._buf, q0._ptr, (3, T, D), (D, 3 * D, 1), q0.dtype)
t0 = bench(lambda: hp.gemm(x, wqkv, blocks))
t1 = bench(lambda: L.call("pdn_qkv_rope_fwd_f32", x._ptr, wqkv._ptr, K * D, q1._ptr, tab._ptr, T, D, K, Lq, hd, K, hp.stream()))
hp.synchronize()
e3 = np.abs(q0.get()[:1024, 2 * D:] - q1.get()[:1024, 2 * D:]).max()
fl = 2.0 * T * 3 * D * K
print(f"q|k|v projection,     {T} tokens: plain {t0:7.1f} us, with RoPE in the store {t1:7.1f} us "
      f"({100 * fl / t1 / 1e-6 / PEAK:4.1f} % of peak)   max |d v| {e3:.1e}", flush=True)

# attention with and without RoPE inside (B * H heads of L = 256, hd 48): what the rotated projection saves there
B, H = T // Lq, D // hd
o, lse, do = hp.empty((B, Lq, H, hd)), hp.empty((B, H, Lq)), hp.from_numpy(rng.standard_normal((B, Lq, H, hd), dtype=np.float32))
dq = hp.empty((T, 3 * D))
ws, wsb = hp.workspace(L.query("pdn_attention_bwd_workspace_bytes", B, H, Lq))
qp, kp, vp = q1._ptr, q1._ptr + 4 * D, q1._ptr + 8 * D
for name, rc, rs, bwd in (("RoPE inside", cos._ptr, sin._ptr, "pdn_attention_bwd_f32"), ("rotated q, k", None, None, "pdn_attention_bwd_rotated_f32")):
    tf = bench(lambda: L.call("pdn_attention_fwd_f32", qp, kp, vp, o._ptr, lse._ptr, B, H, Lq, hd, 3 * D, Lq * 3 * D, D, Lq * D, 1,
                              rc, rs, hp.stream()))
    tb = bench(lambda: L.call(bwd, qp, kp, vp, o._ptr, do._ptr, lse._ptr, dq._ptr, dq._ptr + 4 * D, dq._ptr + 8 * D, B, H, Lq, hd,
                              3 * D, Lq * 3 * D, D, Lq * D, 1, cos._ptr, sin._ptr, ws, wsb, hp.stream()))
    print(f"attention {B * H} heads, {name:12s}: fwd {tf:7.1f} us   bwd {tb:7.1f} us", flush=True)
