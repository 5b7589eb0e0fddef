"""Sweep tile config x split-K for the weight-gradient GEMMs (TN, K = tokens)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp
hp.set_device(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rnd = lambda *s: hp.from_numpy(np.random.default_rng(0).standard_normal(s, dtype=np.float32))
x, h768, g288, g768 = rnd(T, 288), rnd(T, 768), rnd(T, 288), rnd(T, 768)


def bench(A, B, C, iters=10):
    hp.gemm(A, B, C); hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters): hp.gemm(A, B, C)
    return t.ms / iters * 1e3


for name, A, B, cs in [("dW 288x288", x.T, g288, (288, 288)), ("dW 288x768", x.T, g768, (288, 768)),
                       ("dW 768x288", h768.T, g288, (768, 288))]:
    C = hp.empty(cs)
    fl = 2.0 * cs[0] * cs[1] * T
    os.environ.pop("PDN_GEMM_CFG", None)
    us = bench(A, B, C)
    res = [f"auto {us:.0f}us {fl/us/1e6:.0f}TF"]
    if len(sys.argv) > 2 and sys.argv[2] == "auto":
        print(name, res[0], flush=True)
        continue
    for cfg in (0, 2, 3, 4, 1, 10, 11):
        for s in (8, 16, 32, 64):
            os.environ["PDN_GEMM_CFG"] = f"{cfg},{s}"
            us = bench(A, B, C)
            res.append(f"c{cfg}s{s}:{us:.0f}")
    print(name, " ".join(res), flush=True)
