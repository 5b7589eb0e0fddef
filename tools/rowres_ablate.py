"""Row-resident projections ON THE CHUNK KERNEL (csrc/gemm_rowres.hip: `pdn_gemm_rowtile_mode(0)`) with their stores switched off
(PDN_ROWRES_EPI_ABLATE=4, timing only): what the store phases cost -- the measurement the tile-piece kernel of round 5 was
designed from.
usage: python tools/rowres_ablate.py [tokens=65536]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib

hp.set_device(0)
L = _lib.lib()
L.query("pdn_gemm_rowtile_mode", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = 288
rng = np.random.default_rng(0)
x = hp.from_numpy(rng.standard_normal((T, K), dtype=np.float32))


def bench(fn, iters=10):
    for _ in range(3):
        fn()
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters):
            fn()
    return t.ms / iters * 1e3


for N, bt in ((864, 0), (1536, 0), (768, 1), (32000, 0)):
    w = hp.from_numpy((0.05 * rng.standard_normal((N, K) if bt else (K, N))).astype(np.float32))
    c = hp.empty((T, N))
    fn = lambda: L.call("pdn_gemm_rowres_f32", x._ptr, w._ptr, c._ptr, None, None, T, N, K, K, K if bt else N, N, bt, hp.stream())
    out = []
    for ab in ("0", "4"):
        os.environ["PDN_ROWRES_EPI_ABLATE"] = ab
        out.append(bench(fn, 4 if N > 4096 else 20))
    ideal = 2.0 * T * N * K / 157.3e12 * 1e6
    print(f"N={N:6d} {'NT' if bt else 'NN'}: {out[0]:8.1f} us, stores off {out[1]:8.1f} us, MFMA-only {ideal:8.1f} us", flush=True)
os.environ["PDN_ROWRES_EPI_ABLATE"] = "0"
F = 768
dy = x
wd = hp.from_numpy((0.05 * rng.standard_normal((F, K))).astype(np.float32))
gu, dgu = hp.from_numpy(rng.standard_normal((T, 2 * F), dtype=np.float32)), hp.empty((T, 2 * F))
fn = lambda: L.call("pdn_swiglu_bwd_gemm_f32", dy._ptr, wd._ptr, gu._ptr, dgu._ptr, T, F, K, K, hp.stream())
out = []
for ab in ("0", "4"):
    os.environ["PDN_ROWRES_EPI_ABLATE"] = ab
    out.append(bench(fn, 20))
print(f"dh + SwiGLU bwd: {out[0]:8.1f} us, store phase off {out[1]:8.1f} us, MFMA-only {2.0 * T * F * K / 157.3e12 * 1e6:8.1f} us")
