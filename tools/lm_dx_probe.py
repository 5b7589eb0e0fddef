"""lm_head input-gradient product (tokens x 32000) @ (32000 x 288) under different tile configurations
(PDN_GEMM_CFG=<cfg>[,<splits>] override).  usage: python tools/lm_dx_probe.py [tokens]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp
hp.set_device(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
g = hp.empty((T, 32000)); g.fill(0.01)
w = hp.from_numpy(rng.standard_normal((288, 32000), dtype=np.float32))
ref = None
for cfg in (None, "1", "2", "13", "13,2", "8"):
    if cfg is None: os.environ.pop("PDN_GEMM_CFG", None)
    else: os.environ["PDN_GEMM_CFG"] = cfg
    dx = hp.empty((T, 288))
    hp.gemm(g, w.T, dx); hp.synchronize()
    with hp.Timer() as t:
        for _ in range(5): hp.gemm(g, w.T, dx)
    us = t.ms / 5 * 1e3
    out = dx[:64].get()
    if ref is None: ref = out
    print(f"cfg {str(cfg):6s} {us:9.1f} us  {2.0 * T * 288 * 32000 / us / 1e6:6.1f} TFLOP/s  max|diff| vs default {np.abs(out - ref).max():.2e}", flush=True)
