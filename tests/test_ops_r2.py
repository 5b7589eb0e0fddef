"""split / vsplit / hsplit / dsplit (pydynet/core/function.py:14-166), nll_loss (nn/functional.py:353-361)
and float16 operands (the reference's tests draw them: tests/test_tensor_basic.py:16,80-81) against
vectors from the REAL reference (tools/gen_golden_r2.py -> tests/golden/ops_r2.npz)."""
import os

import numpy as np

import pydynet_amd as pdn
import pydynet_amd.nn.functional as F
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants

G = os.path.join(os.path.dirname(__file__), "golden")


def host(x):
    if isinstance(x, pdn.Tensor):
        return x.numpy()
    return x if isinstance(x, np.ndarray) else x.get()


def check_split_family(dev):
    d = np.load(os.path.join(G, "ops_r2.npz"))
    cases = [("split_a0", lambda x: pdn.split(x, 2, axis=0)), ("split_a1", lambda x: pdn.split(x, 3, axis=1)),
             ("split_a2", lambda x: pdn.split(x, 4, axis=2)), ("split_a3", lambda x: pdn.split(x, 2, axis=3)),
             ("split_idx", lambda x: pdn.split(x, (1, 4), axis=1)), ("vsplit", lambda x: pdn.vsplit(x, 2)),
             ("hsplit", lambda x: pdn.hsplit(x, (2, 3))), ("dsplit", lambda x: pdn.dsplit(x, 2))]
    for tag, fn in cases:
        Graph.clear()
        x = pdn.Tensor(d["split/x"], dtype=np.float32, device=dev, requires_grad=True)
        parts = fn(x)
        assert len(parts) == int(d[f"{tag}/n"]), tag
        loss = None
        for i, p in enumerate(parts):
            assert np.array_equal(host(p), d[f"{tag}/{i}"]), (tag, i)          # bit-exact: pure indexing
            term = (p * float(i + 1)).sum()
            loss = term if loss is None else loss + term
        loss.backward()
        assert np.array_equal(host(x.grad), d[f"{tag}/dx"]), tag
    # round trip of the reference's own test (tests/test_ops_extended.py:81-90)
    for axis in (0, 1, 2):
        x = pdn.Tensor(d["split/x"], dtype=np.float32, device=dev)
        assert np.array_equal(host(pdn.concat(pdn.split(x, 2, axis=axis), axis=axis)), d["split/x"])


def check_nll_loss(dev):
    d = np.load(os.path.join(G, "ops_r2.npz"))
    for red in ("mean", "sum"):
        Graph.clear()
        logits = pdn.Tensor(d[f"nll_{red}/logits"], dtype=np.float32, device=dev, requires_grad=True)
        lp = F.log_softmax(logits, axis=1, keepdims=True)
        loss = F.nll_loss(lp, pdn.Tensor(d[f"nll_{red}/onehot"], dtype=np.float32, device=dev), reduction=red)
        assert np.allclose(host(loss), d[f"nll_{red}/loss"], rtol=1e-5)
        loss.backward()
        assert np.allclose(host(logits.grad), d[f"nll_{red}/dlogits"], rtol=1e-4, atol=1e-6)
    import pytest
    with pytest.raises(ValueError):
        F.nll_loss(logits, logits, reduction="max")


def check_float16_operands(dev):
    """float16 storage (results are float16 where the reference's are, computed in float32 on the HIP
    device: every value is within one float16 ulp of NumPy's own float16 arithmetic)."""
    d = np.load(os.path.join(G, "ops_r2.npz"))
    a, b, c = d["f16/a"], d["f16/b"], d["f16/c"]

    def same_kind(got, want, what):
        got = host(got)
        assert got.dtype == want.dtype and got.shape == want.shape, (what, got.dtype, want.dtype)
        g64, w64 = got.astype(np.float64), want.astype(np.float64)
        ok = np.isclose(g64, w64, rtol=2e-3, atol=2e-3) | (np.isnan(g64) & np.isnan(w64)) | (g64 == w64)
        assert ok.all(), (what, g64[~ok][:4], w64[~ok][:4])
    T = lambda arr, rg=False: pdn.Tensor(arr, dtype=arr.dtype, device=dev, requires_grad=rg)
    with np.errstate(all="ignore"):
        for n in ("add", "sub", "mul", "div", "maximum", "minimum"):
            Graph.clear()
            same_kind(getattr(pdn, n)(T(a), T(b)).data, d[f"f16/{n}_hh"], n + " f16,f16")
            same_kind(getattr(pdn, n)(T(a), T(c)).data, d[f"f16/{n}_hs"], n + " f16,f32")
        same_kind(pdn.pow(T(np.abs(a) + np.float16(0.5)), T(b)).data, d["f16/pow_hh"], "pow")
        for n in ("exp", "log", "abs", "sign"):
            same_kind(getattr(pdn, n)(T(np.abs(b) if n == "log" else a)).data, d[f"f16/{n}"], n)
        for n, kw in (("sum", dict(axis=-1)), ("mean", dict(axis=0)), ("max", dict(axis=(0, 2))), ("min", dict())):
            same_kind(np.asarray(host(getattr(pdn, n)(T(a), **kw).data)), d[f"f16/r_{n}"], "reduce " + n)
        same_kind(pdn.matmul(T(b), T(b.T.copy())).data, d["f16/matmul"], "matmul")
        Graph.clear()
        xa, xb = T(a, True), T(b, True)
        y = (xa * xb + pdn.exp(xa)).sum()
        y.backward()
        same_kind(np.asarray(host(y.data)), d["f16/g_out"], "graph out")
        same_kind(xa.grad, d["f16/g_a"], "graph grad a")
        same_kind(xb.grad, d["f16/g_b"], "graph grad b")


for _fn in (check_split_family, check_nll_loss, check_float16_operands):
    device_variants(globals(), _fn)


def test_split_family_cpu():
    check_split_family("cpu")


def test_nll_loss_cpu():
    check_nll_loss("cpu")


def test_float16_operands_cpu():
    check_float16_operands("cpu")
