"""Row-resident projections on the chunk kernel (csrc/gemm_rowres.hip, mode 0) and the tile-piece kernel
(csrc/gemm_rowtile.hip, mode 1): q | k | v + RoPE, gate | up + SwiGLU, dh + SwiGLU backward, plain NN / NT, lm_head + row maxima.
usage: python tools/rowtile_probe.py [tokens=65536]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib

hp.set_device(0)
L = _lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K, F, D, Lq, hd, V = 288, 768, 288, 256, 48, 32000
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
wqkv = stack([(0.05 * rng.standard_normal((K, D))).astype(np.float32) for _ in range(3)])
wv = hp.from_numpy((0.05 * rng.standard_normal((K, V))).astype(np.float32))
gu, h, dgu, qkv = hp.empty((T, 2 * F)), hp.empty((T, F)), hp.empty((T, 2 * F)), hp.empty((T, 3 * D))
c768, c1536 = hp.empty((T, F)), hp.empty((T, 2 * F))
inv = 1.0 / (10000 ** (np.arange(0, hd, 2)[: hd // 2] / hd))
fr = np.outer(np.arange(Lq), inv)
cos, sin = hp.from_numpy(np.cos(fr).astype(np.float32)), hp.from_numpy(np.sin(fr).astype(np.float32))
tab = hp.empty((Lq, hd, 2), np.float32)
L.call("pdn_rope_table_f32", cos._ptr, sin._ptr, tab._ptr, Lq, hd, hp.stream())
L.call("pdn_gateup_swiglu_fwd_f32", x._ptr, wgu._ptr, K * F, gu._ptr, h._ptr, T, F, K, K, hp.stream())
logits = hp.empty((T, V)) if T * V * 4 < 20e9 else None
cases = [
    ("q|k|v + RoPE   (N  864)", 2.0 * T * 3 * D * K, 20,
     lambda: L.call("pdn_qkv_rope_fwd_f32", x._ptr, wqkv._ptr, K * D, qkv._ptr, tab._ptr, T, D, K, Lq, hd, K, hp.stream())),
    ("gate|up + SwiGLU (1536)", 2.0 * T * 2 * F * K, 20,
     lambda: L.call("pdn_gateup_swiglu_fwd_f32", x._ptr, wgu._ptr, K * F, gu._ptr, h._ptr, T, F, K, K, hp.stream())),
    ("dh + SwiGLU bwd  ( 768)", 2.0 * T * F * K, 20,
     lambda: L.call("pdn_swiglu_bwd_gemm_f32", x._ptr, wd._ptr, gu._ptr, dgu._ptr, T, F, K, K, hp.stream())),
    ("plain NN         (1536)", 2.0 * T * 2 * F * K, 20,
     lambda: L.call("pdn_gemm_rowres_f32", x._ptr, wgu._ptr, c1536._ptr, None, None, T, 2 * F, K, K, 2 * F, 2 * F, 0, hp.stream())),
    ("plain NT         ( 768)", 2.0 * T * F * K, 20,
     lambda: L.call("pdn_gemm_rowres_f32", x._ptr, wd._ptr, c768._ptr, None, None, T, F, K, K, K, F, 1, hp.stream())),
]
if logits is not None:
    parts = max(L.query("pdn_linear_rowmax_parts", T, V, K), 1)
    mx = hp.empty((max(parts, 1024), T))
    cases.append(("lm_head + row maxima   ", 2.0 * T * V * K, 6,
                  lambda: L.call("pdn_linear_rowmax_fwd_f32", x._ptr, wv._ptr, None, logits._ptr, mx._ptr, T, V, K, K, V, V, hp.stream())))
    cases.append(("lm_head plain          ", 2.0 * T * V * K, 6,
                  lambda: L.call("pdn_gemm_rowres_f32", x._ptr, wv._ptr, logits._ptr, None, None, T, V, K, K, V, V, 0, hp.stream())))
for name, fl, it, fn in cases:
    out = []
    for mode in (0, 1, 0, 1):
        L.query("pdn_gemm_rowtile_mode", mode)
        out.append(bench(fn, it))
    L.query("pdn_gemm_rowtile_mode", 1)
    old, new = min(out[0], out[2]), min(out[1], out[3])
    print(f"{name} {T} tokens: chunk kernel {old:8.1f} us ({100 * fl / old / 1e-6 / PEAK:4.1f} %), "
          f"tile-piece {new:8.1f} us ({100 * fl / new / 1e-6 / PEAK:4.1f} %)", flush=True)
