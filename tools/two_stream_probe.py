"""Does running the independent dX / dW products of a Linear on two streams overlap one GEMM's store
tail with the other's main loop?  Times N pairs back to back on one stream vs split over two."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib
hp.set_device(0)
L = _lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rnd = lambda *s: hp.from_numpy(np.random.default_rng(0).standard_normal(s, dtype=np.float32))
s2 = ctypes.c_void_p(); L.call("pdn_stream_create", ctypes.byref(s2), 0); s2 = s2.value
s1 = hp.stream()
ev = [ctypes.c_void_p() for _ in range(4)]
for e in ev: L.call("pdn_event_create", ctypes.byref(e), 0)


def run(shapes, two, iters=20):
    x, g, w = shapes["x"], shapes["g"], shapes["w"]
    dx, dw = hp.empty(x.shape), hp.empty(w.shape)
    hp.workspace(1 << 28)

    def once():
        if two:
            L.call("pdn_event_record", ev[0], s1); L.call("pdn_stream_wait_event", s2, ev[0])
            hp.set_stream(s2); hp.gemm(x.T, g, dw, beta=1.0); hp.set_stream(s1)
            hp.gemm(g, w.T, dx)
            L.call("pdn_event_record", ev[1], s2); L.call("pdn_stream_wait_event", s1, ev[1])
        else:
            hp.gemm(x.T, g, dw, beta=1.0)
            hp.gemm(g, w.T, dx)
    for _ in range(3): once()
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(iters): once()
    return t.ms / iters * 1e3


for name, fin, fout in (("288->288", 288, 288), ("288->768", 288, 768), ("768->288", 768, 288), ("288->32000", 288, 32000)):
    sh = {"x": rnd(T, fin), "g": rnd(T, fout) if fout < 4000 else hp.empty((T, fout)), "w": rnd(fin, fout)}
    a, b = run(sh, False), run(sh, True)
    print(f"{name:12s} T={T}: one stream {a:8.1f} us   two streams {b:8.1f} us   ({100 * (a - b) / a:+.1f} %)", flush=True)
