"""Weight-gradient products of the MLP (examples/pydynet/mnist.py:70-78) at batch 65536, x^T @ g with both operands
token-major: time per call and fraction of the fp32-MFMA peak, for each tile shape of the streaming TN kernel
(PDN_GEMM_STREAM_SHAPE: 0 = 3 x 3 tiles of 32, 1 = 5 x 2, 2 = 2 x 5, 3 = 4 x 2, 4 = 2 x 4; unset = the library's choice)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydynet_amd import hipnp as hp, _lib

L = _lib.lib()
hp.set_device(0)
T = 65536
rng = np.random.default_rng(0)
for (fin, fout) in ((784, 1024), (1024, 1024), (768, 768), (512, 2048)):
    x = hp.from_numpy(rng.standard_normal((T, fin)).astype(np.float32))
    g = hp.from_numpy(rng.standard_normal((T, fout)).astype(np.float32))
    dw = hp.empty((fin, fout), np.float32)
    ref = None
    for shape in ("", "0", "1", "2", "3", "4"):
        if shape:
            os.environ["PDN_GEMM_STREAM_SHAPE"] = shape
        else:
            os.environ.pop("PDN_GEMM_STREAM_SHAPE", None)
        for _ in range(3):
            hp.gemm(x.T, g, dw)
        hp.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            hp.gemm(x.T, g, dw)
        hp.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        got = dw.get()
        if ref is None:
            ref = got
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        print(f"dW {fin:5d} x {fout:5d}  shape {shape or 'auto':>4s}: {us:8.1f} us  {2.0 * T * fin * fout / us / 1e6 / 157.3:.3f} of peak   rel diff vs auto {err:.1e}", flush=True)
