"""`examples/pydynet/ts_prediction.py` (GRU(batch_first) -> last hidden -> Linear, MSE, Adam) and
`autograd1d.py` against vectors from the REAL reference running the same definitions
(tests/models_ts_prediction.py, tools/gen_golden.py::gen_ts_prediction).  On the HIP device the GRU
module runs as ONE fused `gru_sequence` node; this pins it in the example's actual usage (batch_first,
`h_state[:, 0, :]`, a None initial state)."""
import os

import numpy as np

import pydynet_amd as pdn
import pydynet_amd.nn as nn
from pydynet_amd.core.tensor import Graph
from pydynet_amd.optim import Adam
from tests import models_ts_prediction as mt
from tests.conftest import device_variants

G = os.path.join(os.path.dirname(__file__), "golden")


def _host(a):
    return a if isinstance(a, np.ndarray) else a.get()


def _run(dev):
    ref = np.load(os.path.join(G, "ts_prediction.npz"))
    Graph.clear()
    np.random.seed(7)
    got = mt.run(pdn, nn, Adam, device=dev, to_host=_host)
    assert np.allclose(got["losses"], ref["losses"], rtol=1e-4), (got["losses"], ref["losses"])
    for k in ref.files:
        if k in ("losses", "autograd1d"):
            continue
        a, r = got[k].astype(np.float64), ref[k].astype(np.float64)
        assert np.linalg.norm(a - r) <= 1e-4 * np.linalg.norm(r) + 1e-7, (k, float(np.linalg.norm(a - r)))
    Graph.clear()
    xs = mt.autograd1d(pdn, device=dev)
    assert np.allclose(xs, ref["autograd1d"], rtol=1e-12, atol=0), (xs[-1], ref["autograd1d"][-1])


def test_ts_prediction_example_cpu():
    _run("cpu")


def check_ts_prediction_example(dev):
    _run(dev)


device_variants(globals(), check_ts_prediction_example)
