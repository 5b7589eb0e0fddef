"""Run a few GEMM shapes repeatedly (for rocprofv3 --pmc runs).  usage: gemm_one.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as _hpsync
from pydynet_amd import hipnp as hp
hp.set_device(0)
it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rnd = lambda *s: hp.from_numpy(np.random.default_rng(0).standard_normal(s, dtype=np.float32))
T = 32768 if len(sys.argv) > 2 else 16384
x, w288, w768, g768, g288 = rnd(T, 288), rnd(288, 288), rnd(288, 768), rnd(T, 768), rnd(T, 288)
cases = [("fwd288", x, w288, hp.empty((T, 288))), ("fwd768", x, w768, hp.empty((T, 768))),
         ("dW288x768", x.T, g768, hp.empty((288, 768))), ("dW288x288", x.T, g288, hp.empty((288, 288)))]
if len(sys.argv) > 2 and sys.argv[2] == "lm":      # the three vocabulary-projection products
    wv = rnd(288, 32000)
    logits = hp.empty((T, 32000))
    cases = [("lm_fwd", x, wv, logits), ("lm_dX", logits, wv.T, hp.empty((T, 288))),
             ("lm_dW", x.T, logits, hp.empty((288, 32000)))]
elif len(sys.argv) > 2:      # only the weight-gradient shapes
    cases = cases[2:]
for name, A, B, C in cases:
    for _ in range(it):
        hp.gemm(A, B, C)
_hpsync.synchronize()
print("done")
