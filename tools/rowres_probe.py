"""Row-resident projection GEMM (csrc/gemm_rowres.hip) against the tiled kernel: results and time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib
hp.set_device(0)
L = _lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
PEAK = 157.3e12


def bench(fn, iters=20):
    for _ in range(3): fn()
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters): fn()
    return t.ms / iters * 1e3


for K, N, trans in ((288, 288, 0), (288, 864, 0), (288, 1536, 0), (288, 768, 0), (288, 288, 1), (288, 768, 1), (288, 1536, 1), (288, 32000, 0), (288, 32000, 1), (288, 160, 1)):
    big = N > 4000 or os.environ.get('NO_EPI') == '1'
    x = hp.from_numpy(rng.standard_normal((T, K), dtype=np.float32))
    w = hp.from_numpy((rng.standard_normal((N, K) if trans else (K, N), dtype=np.float32) * 0.05))
    res = None if big else hp.from_numpy(rng.standard_normal((T, N), dtype=np.float32))
    bias = None if os.environ.get('NO_EPI') == '1' else hp.from_numpy(rng.standard_normal((N,), dtype=np.float32))
    y0, y1 = hp.empty((T, N)), hp.empty((T, N))
    wv = w.T if trans else w
    if not L.query("pdn_gemm_rowres_supported", T, N, K, K, w.shape[1], N, trans):
        print(f"K={K} N={N} trans={trans}: unsupported"); continue
    f0 = lambda: hp.gemm(x, wv, y0, bias=bias, residual=res)
    f1 = lambda: L.call("pdn_gemm_rowres_f32", x._ptr, w._ptr, y1._ptr, bias._ptr if bias is not None else None, res._ptr if res is not None else None,
                        T, N, K, K, w.shape[1], N, trans, hp.stream())
    os.environ["PDN_GEMM_NO_ROWRES"] = "1"
    f0(); f1(); hp.synchronize()
    a, b = y0.get()[:4096], y1.get()[:4096]
    a2, b2 = y0.get()[-300:], y1.get()[-300:]
    err = max(np.abs(a - b).max(), np.abs(a2 - b2).max()) / np.abs(a).max()
    ref = x.get()[:64].astype(np.float64) @ (w.get().T if trans else w.get()).astype(np.float64)
    if bias is not None: ref = ref + bias.get()
    if res is not None: ref = ref + res.get()[:64]
    err64 = np.abs(b[:64] - ref).max() / np.abs(ref).max()
    t0, t1 = bench(f0), bench(f1)
    fl = 2.0 * T * N * K
    print(f"K={K} N={N:5d} trans={trans}: tiled {t0:8.1f} us ({100*fl/t0/1e-6/PEAK:5.1f} %)   row-resident {t1:8.1f} us "
          f"({100*fl/t1/1e-6/PEAK:5.1f} %)   rel diff {err:.2e}  vs float64 {err64:.1e}", flush=True)
