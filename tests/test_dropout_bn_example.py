"""The reference's `examples/pydynet/dropout_bn.py` (plain / Dropout / BatchNorm1d classifiers trained
jointly through ONE backward of l1 + l2 + l3) against vectors from the REAL reference running the same
model definition (tests/models_dropout_bn.py, tools/gen_golden.py::gen_dropout_bn): pins BatchNorm1d
(batch statistics, running averages, eval mode) on the fused colnorm kernels, and Dropout, whose
masks come from the host RNG on every device."""
import os

import numpy as np

import pydynet_amd as pdn
import pydynet_amd.nn as nn
import pydynet_amd.nn.functional as F
from pydynet_amd.core.tensor import Graph
from pydynet_amd.optim import Adam
from tests import models_dropout_bn as md
from tests.conftest import device_variants

G = os.path.join(os.path.dirname(__file__), "golden")


def _host(a):
    return a if isinstance(a, np.ndarray) else a.get()


def _run(dev):
    ref = np.load(os.path.join(G, "dropout_bn.npz"))
    Graph.clear()
    np.random.seed(42)
    got = md.run(pdn, nn, F, Adam, device=dev, to_host=_host)
    assert np.allclose(got["losses"], ref["losses"], rtol=1e-4), (got["losses"], ref["losses"])
    for k in ref.files:
        if k == "losses":
            continue
        a, r = got[k].astype(np.float64), ref[k].astype(np.float64)
        assert a.shape == r.shape, k
        # north-star tolerance, norm-wise (entries at round-off level are not meaningful one by one)
        assert np.linalg.norm(a - r) <= 1e-4 * np.linalg.norm(r) + 1e-6, (k, float(np.linalg.norm(a - r)))


def test_dropout_bn_example_cpu():
    _run("cpu")


def check_dropout_bn_example(dev):
    _run(dev)


device_variants(globals(), check_dropout_bn_example)
