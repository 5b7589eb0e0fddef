"""Time every distinct GEMM of one Llama training step in isolation and weight it by its count per
step: shows where the GEMM time of the step goes (run on the GPU box).  usage: gemm_shapes.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp
hp.set_device(0)
PEAK = 157.3e12
Bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = Bsz * 256
rnd = lambda *s: hp.from_numpy(np.random.default_rng(0).standard_normal(s, dtype=np.float32))


def bench(A, B, C, iters=10, **kw):
    hp.gemm(A, B, C, **kw); hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters): hp.gemm(A, B, C, **kw)
    return t.ms / iters * 1e3


x, h768, g288, g768 = rnd(T, 288), rnd(T, 768), rnd(T, 288), rnd(T, 768)
w288, w768, wd, wv = rnd(288, 288), rnd(288, 768), rnd(768, 288), rnd(288, 32000)
res = rnd(T, 288)
logits = hp.empty((T, 32000))
rows = [
    ("fwd 288->288", 18, x, w288, hp.empty((T, 288)), {}),
    ("fwd 288->288 +res", 6, x, w288, hp.empty((T, 288)), {"residual": res}),
    ("fwd 288->768", 12, x, w768, hp.empty((T, 768)), {}),
    ("fwd 768->288 +res", 6, h768, wd, hp.empty((T, 288)), {"residual": res}),
    ("dX 288<-288 NT", 12, g288, w288.T, hp.empty((T, 288)), {}),
    ("dX 288<-288 NT +fold", 12, g288, w288.T, hp.empty((T, 288)), {"residual": res}),
    ("dX 288<-768 NT", 12, g768, w768.T, hp.empty((T, 288)), {"residual": res}),
    ("dX 768<-288 NT", 6, g288, wd.T, hp.empty((T, 768)), {}),
    ("dW 288x288 TN", 24, x.T, g288, hp.empty((288, 288)), {"beta": 1.0}),
    ("dW 288x768 TN", 12, x.T, g768, hp.empty((288, 768)), {"beta": 1.0}),
    ("dW 768x288 TN", 6, h768.T, g288, hp.empty((768, 288)), {"beta": 1.0}),
    ("lm_head fwd", 1, x, wv, logits, {}),
    ("lm_head dX NT", 1, logits, wv.T, hp.empty((T, 288)), {}),
    ("lm_head dW TN", 1, x.T, logits, hp.empty((288, 32000)), {"beta": 1.0}),
    ("lm_head dW TN +colsum", 0, x.T, logits, hp.empty((288, 32000)),
     {"beta": 1.0, "b_colsum": hp.zeros((32000,), np.float32), "colsum_accumulate": True}),
]
tot = ideal = 0.0
print(f"tokens = {T}")
for name, cnt, A, B, C, kw in rows:
    us = bench(A, B, C, **kw)
    M, K = A.shape; N = B.shape[1]
    fl = 2.0 * M * N * K
    tot += cnt * us; ideal += cnt * fl / PEAK * 1e6
    print(f"{name:22s} x{cnt:2d}  {us:8.1f} us  {100*fl/us/1e-6/PEAK:5.1f}%   step share {cnt*us/1e3:6.2f} ms  (ideal {cnt*fl/PEAK*1e3:5.2f})", flush=True)
print(f"total {tot/1e3:.2f} ms, at peak {ideal/1e3:.2f} ms -> {100*ideal/tot:.1f}%")
