"""The fused conv -> relu -> max_pool entry points (csrc/conv_direct.hip / csrc/conv_quad.hip) timed alone at the LeNet shapes
of BASELINE config 3: forward, data gradient, weight gradient, each with its FLOPs against the fp32-MFMA peak.
usage: python tools/conv_quad_probe.py [batch=4096]   (PDN_CONV_QUAD=0: the conv_direct.hip kernels)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib

hp.set_device(0)
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
PEAK = 157.3e12
rng = np.random.default_rng(0)


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters):
            fn()
    return t.ms / iters * 1e3


for (C, H, W, O) in ((3, 32, 32, 20), (20, 16, 16, 50)):
    x = hp.from_numpy(rng.standard_normal((N, C, H, W), dtype=np.float32))
    w = hp.from_numpy((0.2 * rng.standard_normal((O, C, 3, 3))).astype(np.float32))
    b = hp.from_numpy(rng.standard_normal((O,), dtype=np.float32))
    pooled = hp.empty((N, O, H // 2, W // 2), np.float32)
    mask = hp.empty((N, O, H * W // 32), np.int32)
    dp = hp.from_numpy(rng.standard_normal((N, O, H // 2, W // 2), dtype=np.float32))
    dx = hp.empty((N, C, H, W), np.float32)
    dw, db = hp.empty((O, C, 3, 3), np.float32), hp.empty((O,), np.float32)
    ws, wsb = hp.workspace(L.query("pdn_conv2d_bwd_weight_workspace_bytes", N, C, H, W, O, 3, 1, 1))
    flop = 2.0 * N * H * W * O * C * 9
    st = hp.stream()
    fwd = lambda: L.call("pdn_conv2d_relu_pool_fwd_f32", x._ptr, w._ptr, b._ptr, pooled._ptr, mask._ptr, N, C, H, W, O, 3, 1, 1, st)
    t = bench(fwd)
    print(f"({C},{H},{W})->{O} N={N}: fwd+relu+pool {t:8.1f} us  {100 * flop / t / 1e-6 / PEAK:5.1f} % of fp32-MFMA peak", flush=True)
    if C > 3:
        bwd = lambda: L.call("pdn_conv2d_relu_pool_bwd_data_f32", dp._ptr, mask._ptr, w._ptr, dx._ptr, N, C, H, W, O, 3, 1, 1, st)
        t = bench(bwd)
        print(f"({C},{H},{W})->{O} N={N}: data gradient  {t:8.1f} us  {100 * flop / t / 1e-6 / PEAK:5.1f} %", flush=True)
    wg = lambda: L.call("pdn_conv2d_relu_pool_bwd_weight_f32", x._ptr, dp._ptr, mask._ptr, dw._ptr, db._ptr, 0, N, C, H, W, O,
                        3, 1, 1, ws, wsb, st)
    t = bench(wg)
    print(f"({C},{H},{W})->{O} N={N}: weight gradient {t:7.1f} us  {100 * flop / t / 1e-6 / PEAK:5.1f} %", flush=True)
