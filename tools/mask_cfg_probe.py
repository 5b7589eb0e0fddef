import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from pydynet_amd import hipnp as hp, _lib
L = _lib.lib(); hp.set_device(0)
rng = np.random.default_rng(0)
M = 65536
for (K, N, nt) in ((784, 1024, 0), (1024, 1024, 0), (1024, 1024, 1)):
    x = hp.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    w = hp.from_numpy(rng.standard_normal((N, K) if nt else (K, N)).astype(np.float32))
    b = hp.from_numpy(rng.standard_normal(N).astype(np.float32))
    h = hp.empty((M, N), np.float32); bits = hp.empty((M, N // 32), np.float32)
    for cfg in ("", "0", "7", "8", "3"):
        if cfg: os.environ["PDN_GEMM_CFG"] = cfg
        else: os.environ.pop("PDN_GEMM_CFG", None)
        def run():
            if nt:
                L.call("pdn_linear_dx_masked_f32", x._ptr, K, w._ptr, K, 1, h._ptr, N, None, bits._ptr, None, M, N, K, hp.stream())
            else:
                L.call("pdn_linear_relu_fwd_f32", x._ptr, K, w._ptr, N, 1, b._ptr, h._ptr, N, bits._ptr, M, N, K, hp.stream())
        for _ in range(3): run()
        hp.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run()
        hp.synchronize(); us = (time.perf_counter() - t0) / 10 * 1e6
        print(f"{'dx_masked' if nt else 'relu_fwd'} {M}x{N}x{K} cfg {cfg or 'auto':>4s}: {us:8.1f} us  {2.0*M*N*K/us/1e6/157.3:.3f}", flush=True)
