"""Sweep the tiled kernel's tile configurations (PDN_GEMM_CFG) on one shape.
usage: gemm_cfg_sweep.py M N K [nt]   (nt: B given transposed, the `grad @ W^T` form)"""
import sys, os, subprocess
if len(sys.argv) > 5:      # child: time one configuration
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from pydynet_amd import hipnp as hp
    hp.set_device(0)
    M, N, K = (int(v) for v in sys.argv[1:4]); nt = sys.argv[4] == "1"
    rnd = lambda *s: hp.from_numpy(np.random.default_rng(0).standard_normal(s, dtype=np.float32))
    A, C = rnd(M, K), hp.empty((M, N))
    B = rnd(N, K).T if nt else rnd(K, N)
    os.environ["PDN_GEMM_NO_OUTRES"] = os.environ["PDN_GEMM_NO_ROWRES"] = "1"
    hp.gemm(A, B, C); hp.synchronize()
    with hp.Timer() as t:
        for _ in range(20): hp.gemm(A, B, C)
    us = t.ms / 20 * 1e3
    print(f"cfg {os.environ.get('PDN_GEMM_CFG', 'auto'):6s} {us:8.1f} us  {100 * 2.0 * M * N * K / us / 1e-6 / 157.3e12:5.1f} %")
else:
    M, N, K = sys.argv[1:4]; nt = sys.argv[4] if len(sys.argv) > 4 else "0"
    for cfg in ["auto"] + [str(i) for i in range(13)]:
        env = dict(os.environ)
        if cfg != "auto": env["PDN_GEMM_CFG"] = cfg
        subprocess.run([sys.executable, __file__, M, N, K, nt, "child"], env=env)
