"""Micro-benchmark of pdn_gemm_f32 on the Llama hot-path shapes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp

hp.set_device(0)
PEAK = 157.3e12


def bench(name, A, B, C, iters=20):
    hp.gemm(A, B, C)
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters):
            hp.gemm(A, B, C)
    ms = t.ms / iters
    M, K = A.shape[-2:]; N = B.shape[-1]
    nb = int(np.prod(C.shape[:-2])) if C.ndim > 2 else 1
    fl = 2.0 * M * N * K * nb
    print(f"{name:34s} M={M:6d} N={N:6d} K={K:6d} nb={nb:4d}  {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF/s  {100*fl/ms/1e-3/PEAK:5.1f}% of fp32-MFMA peak", flush=True)


def rnd(*shape):
    return hp.from_numpy(np.random.default_rng(0).standard_normal(shape, dtype=np.float32))


for Bsz in (16, 64):
    T = Bsz * 256
    x, w288, w768, wv = rnd(T, 288), rnd(288, 288), rnd(288, 768), rnd(288, 32000)
    h768, wd = rnd(T, 768), rnd(768, 288)
    g288, g768 = rnd(T, 288), rnd(T, 768)
    print(f"--- tokens = {T}")
    bench("linear 288->288 fwd (NN)", x, w288, hp.empty((T, 288)))
    bench("linear 288->768 fwd (NN)", x, w768, hp.empty((T, 768)))
    bench("linear 768->288 fwd (NN)", h768, wd, hp.empty((T, 288)))
    bench("dX 288<-768 (NT)", g768, w768.T, hp.empty((T, 288)))
    bench("dW 288x768 (TN, split-K)", x.T, g768, hp.empty((288, 768)))
    bench("dW 288x288 (TN, split-K)", x.T, g288, hp.empty((288, 288)))
    logits = hp.empty((T, 32000))
    bench("lm_head fwd (NN)", x, wv, logits)
    bench("lm_head dX (NT)", logits, wv.T, hp.empty((T, 288)))
    bench("lm_head dW (TN)", x.T, logits, hp.empty((288, 32000)))
    del logits
    q = rnd(Bsz, 256, 6, 48); k = rnd(Bsz, 256, 6, 48)
    s = hp.empty((Bsz, 6, 256, 256))
    bench("attn QK^T (batched NT views)", q.transpose(0, 2, 1, 3), k.transpose(0, 2, 3, 1), s)
    bench("attn PV (batched NN views)", s, k.transpose(0, 2, 1, 3), hp.empty((Bsz, 6, 256, 48)))

# ---- tile-shape sweep on the key shapes (PDN_GEMM_CFG override) -------------------------------
import os
print("--- sweep: cfg 0=128x128 1=128x96 2=96x128 3=128x64 4=64x128 5=64x64 6=256x128 7=128x256 8=256x96 9=96x256 10=64x96 11=96x64")
T = 64 * 256
x, w288, w768, wv = rnd(T, 288), rnd(288, 288), rnd(288, 768), rnd(288, 32000)
g768, g288, h768, wd = rnd(T, 768), rnd(T, 288), rnd(T, 768), rnd(768, 288)
logits = rnd(T // 4, 32000)
shapes = [("fwd 288->288", x, w288, (T, 288), "1 3 8 10"), ("fwd 288->768", x, w768, (T, 768), "0 1 6 7 8"),
          ("fwd 768->288", h768, wd, (T, 288), "1 8 10"), ("dX 768->288 NT", g768, w768.T, (T, 288), "1 8 10"),
          ("dX 288->768 NT", g288, wd.T, (T, 768), "0 1 6 7 8"),
          ("dW 288x768 TN", x.T, g768, (288, 768), "2,4 2,8 4,4 4,8 9,4 9,8 9,16 11,8 11,16 0,8"),
          ("dW 288x288 TN", x.T, g288, (288, 288), "2,8 2,16 4,8 4,16 9,8 9,16 11,8 11,16 11,32"),
          ("dW 768x288 TN", h768.T, g288, (768, 288), "2,4 2,8 3,8 9,8 11,8 11,16 0,8"),
          ("lm_head fwd", x[:T // 4], wv, (T // 4, 32000), "0 6 7"),
          ("lm_head dX NT", logits, wv.T, (T // 4, 288), "1 8 10"),
          ("lm_head dW TN", x[:T // 4].T, logits, (288, 32000), "2 9 0")]
for name, A, B, cs, cfgs in shapes:
    C = hp.empty(cs)
    for cfg in cfgs.split():
        os.environ["PDN_GEMM_CFG"] = cfg
        bench(f"{name} cfg={cfg}", A, B, C, iters=10)
    os.environ.pop("PDN_GEMM_CFG")
    bench(f"{name} auto", A, B, C, iters=10)
