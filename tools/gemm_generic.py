"""pdn_gemm_f32 on GENERIC shapes (not the 288-wide Llama ones): square, MLP (784 / 1024), dim-512 / 768
transformer blocks at 65536 tokens -- forward (NN), input gradient (NT), weight gradient (TN).  Shows what the
tiled / streaming kernels reach where no resident-operand kernel applies.  usage: gemm_generic.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp
hp.set_device(0)
PEAK = 157.3e12
rnd = lambda *s: hp.from_numpy(np.random.default_rng(0).standard_normal(s, dtype=np.float32))


def bench(name, A, B, C, iters=10):
    hp.gemm(A, B, C); hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters):
            hp.gemm(A, B, C)
    us = t.ms / iters * 1e3
    M, K = A.shape; N = B.shape[1]
    fl = 2.0 * M * N * K
    print(f"{name:28s} M={M:6d} N={N:6d} K={K:6d}  {us:9.1f} us  {100 * fl / us / 1e-6 / PEAK:5.1f} %", flush=True)


for n in (2048, 4096, 8192):
    bench(f"square {n}", rnd(n, n), rnd(n, n), hp.empty((n, n)))
T = 65536
for D, F in ((512, 2048), (768, 3072), (1024, 1024), (784, 1024)):
    x, w, g = rnd(T, D), rnd(D, F), rnd(T, F)
    bench(f"fwd  {D}->{F} (NN)", x, w, hp.empty((T, F)))
    bench(f"dX   {D}<-{F} (NT)", g, w.T, hp.empty((T, D)))
    bench(f"dW   {D}x{F} (TN)", x.T, g, hp.empty((D, F)))
    if D != F:
        w2 = rnd(F, D)
        bench(f"fwd  {F}->{D} (NN)", g, w2, hp.empty((T, D)))
    wd = rnd(D, D)
    bench(f"fwd  {D}->{D} (NN)", x, wd, hp.empty((T, D)))
