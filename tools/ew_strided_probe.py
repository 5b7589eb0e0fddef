"""Strided / broadcasting elementwise kernels at the shapes the plain-operator Llama (tests/models_plain_llama.py) produces:
RoPE's `even * cos` on a stride-2 view against a broadcast table, the transposed copy in front of the score product, a
broadcast add.  Prints us per launch and the HBM rate of the bytes each one has to touch (sectors, not elements).
usage: python tools/ew_strided_probe.py [batch=256]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp

hp.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, H, HD = 256, 6, 48
rng = np.random.default_rng(0)
x = hp.from_numpy(rng.standard_normal((B, L, H, HD // 2, 2), dtype=np.float32))
cos = hp.from_numpy(rng.standard_normal((L, 1, HD // 2), dtype=np.float32))
y = hp.from_numpy(rng.standard_normal((B, L, H, HD), dtype=np.float32))
bias = hp.from_numpy(rng.standard_normal((HD,), dtype=np.float32))


def t(f, n=20):
    f(); hp.synchronize()
    with hp.Timer() as tm:
        for _ in range(n):
            f()
    return tm.ms / n * 1e3


even = x[..., 0]
cases = {
    "even * cos   (stride-2 view x broadcast table)": (lambda: even * cos, 4.0 * x.size + 4.0 * even.size),
    "transposed copy (B, L, H, hd) -> (B, H, L, hd)": (lambda: hp.ascontiguousarray(y.transpose(0, 2, 1, 3)), 8.0 * y.size),
    "row-broadcast add (B, L, H, hd) + (hd,)": (lambda: y + bias, 8.0 * y.size),
    "contiguous add (vector path, for scale)": (lambda: y + y, 12.0 * y.size),
}
for name, (f, nbytes) in cases.items():
    us = t(f)
    print(f"{name:50s} {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s")
ref = np.asarray(even.get()) * np.asarray(cos.get())
assert np.array_equal(np.asarray((even * cos).get()), ref)
assert np.array_equal(np.asarray(hp.ascontiguousarray(y.transpose(0, 2, 1, 3)).get()), np.ascontiguousarray(np.asarray(y.get()).transpose(0, 2, 1, 3)))
print("values equal NumPy's")
