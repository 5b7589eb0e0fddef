"""Output-resident GEMM (csrc/gemm_outres.hip, N = 288) against the tiled kernel: results and time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib
hp.set_device(0)
L = _lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
PEAK, N = 157.3e12, 288
os.environ["PDN_GEMM_NO_OUTRES"] = "1"


def bench(fn, iters=20):
    for _ in range(3): fn()
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters): fn()
    return t.ms / iters * 1e3


for K, trans, extras in ((288, 0, 0), (288, 1, 0), (768, 0, 0), (768, 0, 2), (864, 1, 0), (864, 1, 2), (1536, 1, 2), (32000, 1, 0), (32000, 0, 0)):
    big = K > 4000
    x = hp.empty((T, K)) if big else hp.from_numpy(rng.standard_normal((T, K), dtype=np.float32))
    if big:
        x[...] = 0.01
        x[:4096] = hp.from_numpy(rng.standard_normal((4096, K), dtype=np.float32))
    w = hp.from_numpy((rng.standard_normal((N, K) if trans else (K, N), dtype=np.float32) * 0.05))
    res = hp.from_numpy(rng.standard_normal((T, N), dtype=np.float32)) if extras & 2 else None
    y0, y1 = hp.empty((T, N)), hp.empty((T, N))
    wv = w.T if trans else w
    f0 = lambda: hp.gemm(x, wv, y0, residual=res)
    f1 = lambda: L.call("pdn_gemm_outres_f32", x._ptr, w._ptr, y1._ptr, None, res._ptr if res is not None else None,
                        T, N, K, K, w.shape[1], N, trans, hp.stream())
    f0(); f1(); hp.synchronize()
    a, b = y0.get()[:4096], y1.get()[:4096]
    a2, b2 = y0.get()[-300:], y1.get()[-300:]
    err = max(np.abs(a - b).max(), np.abs(a2 - b2).max()) / np.abs(a).max()
    ref = x[:64].get().astype(np.float64) @ (w.get().T if trans else w.get()).astype(np.float64)
    if res is not None: ref = ref + res.get()[:64]
    err64 = np.abs(b[:64] - ref).max() / np.abs(ref).max()
    t0, t1 = bench(f0, 5 if big else 20), bench(f1, 5 if big else 20)
    fl = 2.0 * T * N * K
    print(f"K={K:5d} trans={trans} res={extras >> 1}: tiled {t0:8.1f} us ({100*fl/t0/1e-6/PEAK:5.1f} %)   output-resident {t1:8.1f} us "
          f"({100*fl/t1/1e-6/PEAK:5.1f} %)   rel diff {err:.2e}  vs float64 {err64:.1e}", flush=True)
