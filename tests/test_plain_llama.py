"""The plain-operator Llama (tests/models_plain_llama.py: RoPE as slices + concat, attention as matmul -> softmax -> matmul,
the reference's own formulation llm/llama/model.py:23-44, 95-121) pinned to the vectors the REAL reference produced for
the tiny model (tests/golden/tiny_llama.npz, tools/gen_golden.py): five losses, first-step gradients, final parameters.
Runs on "cpu", on the emulated C ABI and (-m gpu) on a real MI355X."""
import os

import numpy as np

from pydynet_amd.core.tensor import Graph
from pydynet_amd.optim import Adam
from tests.conftest import device_variants
from tests.models_plain_llama import PlainLlama

G = os.path.join(os.path.dirname(__file__), "golden")
RT = 1e-4


def _host(a):
    return a if isinstance(a, np.ndarray) else a.get()


def check_plain_llama_five_steps(dev):
    d = np.load(os.path.join(G, "tiny_llama.npz"))
    np.random.seed(1234)
    m = PlainLlama(64, 48, 2, 96, 64, 2, np.float32)
    names = [k[5:] for k in d.files if k.startswith("init/")]
    assert sorted(names) == sorted(n for n, _ in m.named_parameters())
    for n in names:
        m._parameters[n].data[...] = d["init/" + n]
    m.to(dev)
    opt = Adam(m.parameters(), lr=1e-3)
    losses = []
    for s in range(5):
        losses.append(m.finetune_step(d["ids"], d["tgt"], opt))
        if s == 0:
            for n in names:
                g, want = _host(m._parameters[n].grad), d["grad1/" + n]
                scale = max(float(np.abs(want).max()), 1e-30)
                assert float(np.abs(g - want).max()) <= 1e-7 + RT * scale, n
    assert np.allclose(losses, d["losses"], rtol=RT, atol=0), (losses, d["losses"])
    for n in names:
        a, b = _host(m._parameters[n].data), d["final/" + n]
        err = np.abs(a.astype(np.float64) - b)
        bad = err > 1e-6 + RT * float(np.abs(b).max())
        assert bad.sum() <= max(1, a.size // 500) and float(err.max()) <= 2 * 1e-3 * 5, (n, int(bad.sum()))


def test_plain_llama_five_steps_cpu():
    Graph.clear()
    check_plain_llama_five_steps("cpu")


device_variants(globals(), check_plain_llama_five_steps)
