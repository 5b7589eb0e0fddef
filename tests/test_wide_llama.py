"""(Also, at the end: the benchmarked width at 512 positions, `long288_llama.npz`.)
A Llama of ANOTHER width than the benchmarked 288 (the reference's constructor is general: llm/llama/model.py:153-197)
against the REAL reference: `tests/golden/wide_llama.npz` holds the loss, the norm and a strided sample of every gradient
of one training step of a one-layer model of width 512 (head dim 64, ffn 1376) over 4096 tokens, produced by
`tools/gen_golden_r2.py wide_llama` importing /root/reference.  At this width the row-resident kernels do not apply: the
step must take the SwiGLU epilogues of the TILED kernel (round 5, csrc/gemm.hip SWI; `ffn_swiglu.tiled_epilogue`), which the
library's launch counters confirm.  Tolerance 1e-4 relative to each gradient's largest entry (north_star)."""
import ctypes
import os

import numpy as np
import pytest

import pydynet_amd as pdn
from pydynet_amd.llm.llama import Llama
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASE = dict(V=256, D=512, H=8, F=1376, L=256, B=16, seed=11)          # = tools/gen_golden_r2.py WIDE_CASE
LONG288 = dict(V=64, D=288, H=6, F=768, L=512, B=8, seed=12)          # = tools/gen_golden_r2.py LONG288_CASE


def _step(dev, c=CASE):
    Graph.clear()
    np.random.seed(c["seed"])
    m = Llama(c["V"], c["D"], c["H"], c["F"], c["L"], c["B"], 1, np.float32)
    m.tok_embedding.weight.data[...] = (0.05 * np.random.randn(c["V"], c["D"])).astype(np.float32)
    m = m.to(dev)
    rng = np.random.default_rng(c["seed"])
    ids, tgt = rng.integers(0, c["V"], (c["B"], c["L"])), rng.integers(0, c["V"], (c["B"], c["L"]))
    m.train(True)
    loss = m.loss(ids, tgt)
    loss.backward()
    return float(loss.item()), {n: p.grad.get() if hasattr(p.grad, "get") else np.array(p.grad) for n, p in m.named_parameters()}


def _check(dev, fixture="wide_llama.npz", case=CASE, want=None):
    ref = np.load(os.path.join(G, fixture))
    if dev != "cpu":
        from pydynet_amd import _lib
        buf = (ctypes.c_int64 * 21)()
        _lib.lib().call("pdn_kernel_counters", buf, 21, 1)
    loss, grads = _step(dev, case)
    if dev != "cpu":
        _lib.lib().call("pdn_kernel_counters", buf, 21, 1)
        if want is None:
            assert buf[19] == 1 and buf[20] == 1, ("SwiGLU in the tiled kernel's stores (forward, backward)", buf[19], buf[20])
        else:
            want(list(buf))
    ref_loss = float(ref["loss"])
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss), (loss, ref_loss)
    names = [k[6:] for k in ref.files if k.startswith("gnorm/")]
    assert sorted(names) == sorted(grads), (sorted(names), sorted(grads))
    for n in names:
        g = grads[n].astype(np.float64).reshape(-1)
        gn, gmax = float(ref["gnorm/" + n]), float(ref["gmax/" + n])
        assert abs(float(np.linalg.norm(g)) - gn) <= 1e-4 * gn + 1e-12, (n, float(np.linalg.norm(g)), gn)
        err = float(np.abs(g[::61][:4096] - ref["gsample/" + n]).max())
        assert err <= 1e-4 * gmax + 1e-7, (n, err, gmax)


def check_wide_llama_step(dev):
    _check(dev)


device_variants(globals(), check_wide_llama_step)


def test_wide_llama_step_cpu():
    _check("cpu")


# ---- width 288 at 512 positions: RoPE in the projection's store, attention as 256-row block pairs (round 5) ------------
def _long288_kernels(c):
    # one layer: the persistent attention kernels once per (query block, key block <= query block) pair on the GPU (the
    # emulator counts a call once), no resident / streaming attention launch; the rotated projection
    assert c[7] in (1, 3) and c[8] in (1, 3) and c[9] == 0 and c[10] == 0 and c[11] == 0, ("attention launches", c[7:12])
    assert c[4] + c[6] >= 1, ("q | k | v projection with RoPE in its store", c[4], c[6])


def check_long288_llama_step(dev):
    """`tests/golden/long288_llama.npz` (tools/gen_golden_r2.py long288_llama, the real reference): one step of a one-layer
    Llama of width 288 / head dim 48 at 512 positions over 4096 tokens."""
    _check(dev, "long288_llama.npz", LONG288, _long288_kernels)


device_variants(globals(), check_long288_llama_step)


def test_long288_llama_step_cpu():
    _check("cpu", "long288_llama.npz", LONG288)
