"""pooling in one dimension, LSTM, RNN, SGD (momentum / nesterov / weight decay),
Adagrad, Adadelta and the learning-rate schedulers against vectors from the REAL reference running the
same scenario (tests/models_misc.py, tools/gen_golden.py::gen_misc).  (The reference's own Conv1d and
BatchNorm2d cannot be executed -- see the notes in models_misc.py -- so they have nothing to pin.)"""
import os

import numpy as np

import pydynet_amd as pdn
import pydynet_amd.nn as nn
import pydynet_amd.nn.functional as F
import pydynet_amd.optim as optim
from pydynet_amd.core.tensor import Graph
from pydynet_amd.optim import lr_scheduler
from tests import models_misc as mm
from tests.conftest import device_variants

G = os.path.join(os.path.dirname(__file__), "golden")


def _host(a):
    return a if isinstance(a, np.ndarray) else a.get()


def _run(dev):
    ref = np.load(os.path.join(G, "misc_layers.npz"))
    Graph.clear()
    got = mm.run(pdn, nn, F, optim, lr_scheduler, device=dev, to_host=_host)
    assert set(got) == set(ref.files)
    for k in ref.files:
        a, r = np.asarray(got[k], np.float64), ref[k].astype(np.float64)
        assert a.shape == r.shape, (k, a.shape, r.shape)
        if k.startswith("lr/"):
            assert np.allclose(a, r, rtol=1e-12, atol=1e-15), k           # host arithmetic: exact schedule
        else:
            assert np.linalg.norm(a - r) <= 1e-4 * np.linalg.norm(r) + 1e-6, (k, float(np.linalg.norm(a - r)))


def test_misc_layers_cpu():
    _run("cpu")


def check_misc_layers(dev):
    _run(dev)


device_variants(globals(), check_misc_layers)
