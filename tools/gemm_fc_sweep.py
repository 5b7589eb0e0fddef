"""Tile configuration x split-K sweep of the tiled GEMM (PDN_GEMM_CFG="<cfg>,<splits>") on the three products of a Linear
layer: forward x W (NN), input gradient g W^T (NT), weight gradient x^T g (TN).
usage: python tools/gemm_fc_sweep.py rows in_features out_features"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp

hp.set_device(0)
R, I, O = (int(v) for v in sys.argv[1:4])
rnd = lambda *s: hp.from_numpy(np.random.default_rng(0).standard_normal(s, dtype=np.float32))
x, w, g = rnd(R, I), rnd(I, O), rnd(R, O)
cases = {"fwd  x W   ": (x, w, hp.empty((R, O)), 2.0 * R * I * O), "dX   g W^T ": (g, w.T, hp.empty((R, I)), 2.0 * R * I * O),
         "dW   x^T g ": (x.T, g, hp.empty((I, O)), 2.0 * R * I * O)}
REAL = bool(os.environ.get("SWEEP_REAL"))      # SWEEP_REAL=1: only the library's own choice, every kernel family allowed
if not REAL:
    os.environ["PDN_GEMM_NO_OUTRES"] = os.environ["PDN_GEMM_NO_ROWRES"] = os.environ["PDN_GEMM_NO_STREAM"] = "1"


def t(A, B, C, n=10):
    hp.gemm(A, B, C); hp.synchronize()
    with hp.Timer() as tm:
        for _ in range(n):
            hp.gemm(A, B, C)
    return tm.ms / n * 1e3


for name, (A, B, C, fl) in cases.items():
    os.environ.pop("PDN_GEMM_CFG", None)
    res = [("auto", t(A, B, C))]
    for c in range(0 if REAL else 13):
        for s in (1, 2, 4, 8):
            os.environ["PDN_GEMM_CFG"] = f"{c},{s}"
            try:
                res.append((f"{c},{s}", t(A, B, C)))
            except Exception as e:
                pass
    os.environ.pop("PDN_GEMM_CFG", None)
    res[0] = ("auto", min(res[0][1], t(A, B, C)))            # (the first measurement of a process runs on cold clocks)
    if os.environ.get("SWEEP_SHOW"):
        os.environ["PDN_GEMM_DEBUG"] = "1"; hp.gemm(A, B, C); hp.synchronize(); os.environ.pop("PDN_GEMM_DEBUG")
    if not REAL and name.startswith("dW"):                      # the wave-streaming kernel on warm clocks, beside the tiled ones
        os.environ.pop("PDN_GEMM_NO_STREAM"); os.environ["PDN_GEMM_STREAM_MAX"] = str(1 << 40)
        res.append(("stream", t(A, B, C)))
        os.environ["PDN_GEMM_NO_STREAM"] = "1"; os.environ.pop("PDN_GEMM_STREAM_MAX")
    five = "  ".join(f"{k} {u:6.1f}" for k, u in res if k.startswith("5,") or k == "stream")
    res.sort(key=lambda r: r[1])
    auto = [r for r in res if r[0] == "auto"][0][1]
    print(name, f"auto {auto:7.1f} us ({100 * fl / auto / 1e-6 / 157.3e12:4.1f} %)  best:",
          "  ".join(f"{k} {u:6.1f} ({100 * fl / u / 1e-6 / 157.3e12:4.1f} %)" for k, u in res[:6]), "|", five, flush=True)
