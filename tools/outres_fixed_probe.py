"""Fixed cost of an output-resident launch (csrc/gemm_outres.hip): time against the contraction length, with and without the
epilogue (PDN_OUTRES_RT_ABLATE=1: nothing stored) -- intercept = what a launch costs besides its k-pieces.
usage: python tools/outres_fixed_probe.py [tokens=65536]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib

hp.set_device(0)
L = _lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = 288
rng = np.random.default_rng(0)


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters):
            fn()
    return t.ms / iters * 1e3


Kmax = 1536
x = hp.from_numpy(rng.standard_normal((T, Kmax), dtype=np.float32))
y = hp.empty((T, N))
for trans in (0, 1):
    rows = []
    for K in (32, 64, 128, 256, 512, 768, 1024, 1536):
        w = hp.from_numpy((rng.standard_normal((N, K) if trans else (K, N), dtype=np.float32) * 0.05))
        fn = lambda: L.call("pdn_gemm_outres_f32", x._ptr, w._ptr, y._ptr, None, None, T, N, K, Kmax, w.shape[1], N, trans, hp.stream())
        ts = []
        for ab in ("0", "1"):
            os.environ["PDN_OUTRES_RT_ABLATE"] = ab
            ts.append(bench(fn))
        rows.append((K, ts[0], ts[1]))
        print(f"{'NT' if trans else 'NN'} K={K:5d}: {ts[0]:7.1f} us, no epilogue {ts[1]:7.1f} us, MFMA-only {2.0 * T * N * K / 157.3e12 * 1e6:6.1f} us", flush=True)
    ks = np.array([r[0] for r in rows[3:]], float)
    for col, name in ((1, "with epilogue"), (2, "no epilogue")):
        t = np.array([r[col] for r in rows[3:]])
        a, b = np.polyfit(ks, t, 1)
        print(f"   {name}: {a * 32:.3f} us per 32-k piece (MFMA-only {2.0 * T * N * 32 / 157.3e12 * 1e6:.3f}), intercept {b:.1f} us")
os.environ["PDN_OUTRES_RT_ABLATE"] = "0"
