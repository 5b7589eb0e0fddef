"""Where a decode step's time goes: back-to-back graph replays (pure GPU time per token), the generate loop with and
without the step queued ahead, and with the token read through a view (infer.py:55) or whole."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp
import pydynet_amd as pdn
from pydynet_amd.llm.llama import Llama
np.random.seed(0)
model = Llama(32000, 288, 6, 768, 1024, 1, 6, np.float32)
model.tok_embedding.weight.data[...] = (0.02 * np.random.randn(32000, 288)).astype(np.float32)
model = model.to("cuda"); model.eval()
ids = np.random.randint(0, 32000, (1, 8))
N = 264


def run(ahead, view):
    Llama.decode_ahead = ahead
    with pdn.no_grad():
        n, t0 = 0, None
        for tok in model.generate(ids, N):
            _ = (tok[0] if view else tok).numpy()
            n += 1
            if n == 1:
                hp.synchronize(); t0 = time.perf_counter()
        hp.synchronize()
        return (n - 1) / (time.perf_counter() - t0)


with pdn.no_grad():
    run(True, True)
    for ahead in (False, True):
        for view in (False, True):
            print(f"generate ahead={ahead} read_view={view}: {run(ahead, view):7.0f} tok/s", flush=True)
    st = model._decode_st
    g = st["graphs"][1]
    st["pos"][...] = np.int32(20)
    hp.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    hp.synchronize()
    print(f"back-to-back replays: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per token ({g.nodes} nodes)")
    st["host_pos"] = None
