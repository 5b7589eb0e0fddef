"""lm_head forward forms (csrc/gemm_rowres.hip): plain product, product + log-sum-exp (EPI 4), product + row maxima (EPI 5),
the latter also with its maximum switched off (PDN_ROWRES_EPI_ABLATE=2) to see what the butterfly costs.
usage: python tools/lmhead_probe.py [tokens=65536] [vocab=32000]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib

hp.set_device(0)
L = _lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
V = int(sys.argv[2]) if len(sys.argv) > 2 else 32000
K = 288
rng = np.random.default_rng(0)
x = hp.from_numpy(rng.standard_normal((T, K), dtype=np.float32))
w = hp.from_numpy((0.05 * rng.standard_normal((K, V))).astype(np.float32))
logits, st = hp.empty((T, V)), hp.empty((T,))


def bench(fn, iters=8):
    for _ in range(2):
        fn()
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters):
            fn()
    return t.ms / iters * 1e3


fl = 2.0 * T * V * K
rows = [("plain", lambda: hp.gemm(x, w, logits), ""),
        ("+ lse (EPI 4)", lambda: L.call("pdn_linear_lse_fwd_f32", x._ptr, w._ptr, None, logits._ptr, st._ptr, T, V, K, K, V, V, hp.stream()), ""),
        ("+ max (EPI 5)", lambda: L.call("pdn_linear_rowmax_fwd_f32", x._ptr, w._ptr, None, logits._ptr, st._ptr, T, V, K, K, V, V, hp.stream()), "")]
rows.append(("+ max, maximum left out", rows[2][1], "2"))
for name, fn, ab in rows:
    os.environ["PDN_ROWRES_EPI_ABLATE"] = ab or "0"
    us = bench(fn)
    print(f"{name:24s} {us:9.1f} us   {100 * fl / us / 1e-6 / 157.3e12:5.1f} % of the fp32 MFMA peak", flush=True)
