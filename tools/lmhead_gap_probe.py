"""Why the lm_head forward measures slower inside the step than alone: the product is timed
(a) back to back with itself, (b) right after the lm_head weight-gradient product of the previous step (what precedes it
in the training step, modulo the short layer kernels: 8.4 GB of logits read, L2 / MALL full of logits, clocks settled
under a different instruction mix), (c) after a 0.3 s idle gap (clocks dropped).  Events bracket ONLY the forward.
usage: python tools/lmhead_gap_probe.py [tokens=65536]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib

hp.set_device(0)
L = _lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
V, K = 32000, 288
rng = np.random.default_rng(0)
x = hp.from_numpy(rng.standard_normal((T, K), dtype=np.float32))
w = hp.from_numpy((0.05 * rng.standard_normal((K, V))).astype(np.float32))
logits, dw = hp.empty((T, V)), hp.empty((K, V))
parts = max(L.query("pdn_linear_rowmax_parts", T, V, K), 1)
mx = hp.empty((parts, T))
fwd = lambda: L.call("pdn_linear_rowmax_fwd_f32", x._ptr, w._ptr, None, logits._ptr, mx._ptr, T, V, K, K, V, V, hp.stream())
dwf = lambda: hp.gemm(x.T, logits, dw)
small = hp.empty((T, K))
fill = lambda: small.__setitem__(Ellipsis, 1.0)


def timed(before, n=6):
    out = []
    for _ in range(n):
        before()
        with hp.Timer() as t:
            fwd()
        out.append(t.ms * 1e3)
    return np.median(out), min(out), max(out)


for _ in range(3):
    fwd()
hp.synchronize()
fl = 2.0 * T * V * K
for name, before in (("back to back with itself", fwd), ("right after the lm_head weight gradient", dwf),
                     ("after 40 short HBM kernels (75 MB fills)", lambda: [fill() for _ in range(40)]),
                     ("after 0.3 s of idle GPU", lambda: (hp.synchronize(), time.sleep(0.3)))):
    med, lo, hi = timed(before)
    print(f"lm_head forward + row maxima, {name:42s}: median {med:8.1f} us  [{lo:8.1f} .. {hi:8.1f}]  "
          f"{100 * fl / med / 1e-6 / 157.3e12:5.1f} % of peak", flush=True)
