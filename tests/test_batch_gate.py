"""The kernels of the BENCHMARKED batch, end to end.

The full-size Llama is pinned to the real reference at one sequence (tests/test_llama_golden.py), but at 256 tokens
`pdn_gemm_f32` never reaches the row-resident / output-resident kernels that carry ~78 % of the timed step (they
start at 8192 / 57089 rows).  Samples are independent and the loss is a mean over tokens
(llm/llama/model.py:226-252), so ONE finetune step on B sequences must equal the mean of the B single-sequence
steps: `bench.batch_gate` compares the loss and all 58 gradient tensors (1e-4 of each tensor's largest entry; the
embedding's scatter-ASSIGN restated from the per-sequence gradients in batch order, tensor.py:937-940) and asserts
through `pdn_gemm_prof_collect_families` that kernel families 2, 3 and 4 were really launched.  bench.py runs the
same check on its timed inputs before it prints a number."""
import numpy as np
import pytest


def _fullsize(dev):
    from pydynet_amd.llm.llama import Llama
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    np.random.seed(0)
    m = Llama(32000, 288, 6, 768, 1024, 1, 6, np.float32)
    m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(32000, 288)).astype(np.float32)
    return m.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [224, 256, 512])     # 512 = bench.py's default per-GPU batch (round 5), 256 = rounds 1-4's
def test_benchmarked_batch_equals_mean_of_single_sequence_steps_gpu(hip, B):
    import bench
    import pydynet_amd as pdn
    from pydynet_amd import _lib
    m = _fullsize("hip:0")
    rng = np.random.default_rng(1000)                       # bench.py's rank-0 inputs
    ids, tgt = rng.integers(0, 32000, (B, 256)), rng.integers(0, 32000, (B * 256,))
    rec = bench.batch_gate(m, ids, tgt, "hip:0", pdn, _lib.lib(), rtol=1e-4, want_families=(2, 3, 4))
    assert rec["grad_tensors_checked"] == 58 and rec["worst_grad_rel_err"] <= 1e-4 and rec["loss_rel_err"] <= 1e-4
    fam = rec["gemm_family_launches"]
    assert fam["rowres"] > 0 and fam["outres"] > 0 and fam["outres_tn"] > 0


def check_batch_gate_logic(device):
    """The gate itself on a tiny model (duplicate token ids inside and across sequences): passes on a correct
    step, and FAILS when one gradient is perturbed."""
    import bench
    import pydynet_amd as pdn
    from pydynet_amd import _lib
    from pydynet_amd.llm.llama import Llama
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    np.random.seed(3)
    m = Llama(40, 48, 2, 96, 32, 1, 2, np.float32)
    m.tok_embedding.weight.data[...] = (0.05 * np.random.randn(40, 48)).astype(np.float32)
    m = m.to(device)
    rng = np.random.default_rng(5)
    ids, tgt = rng.integers(0, 12, (5, 16)), rng.integers(0, 40, (5 * 16,))
    rec = bench.batch_gate(m, ids, tgt, device, pdn, _lib.lib(), rtol=1e-4, want_families=())
    assert rec["batch"] == 5 and rec["worst_grad_rel_err"] <= 1e-4
    assert rec["grad_tensors_checked"] == len(list(m.named_parameters()))
    # a wrong batched gradient must be caught: make the batched loss see a different target than the per-sequence runs
    orig = m.loss
    calls = {"n": 0}

    def skewed(i, t, *a, **k):
        calls["n"] += 1
        if calls["n"] == 1:                                 # the batched step only
            t = (np.asarray(t) + 1) % 40
        return orig(i, t, *a, **k)
    m.loss = skewed
    with pytest.raises(SystemExit, match="batch gate FAILED"):
        bench.batch_gate(m, ids, tgt, device, pdn, _lib.lib(), rtol=1e-4, want_families=())
    m.loss = orig
    # and a kernel family that did not run is reported
    with pytest.raises(SystemExit, match="families"):
        bench.batch_gate(m, ids, tgt, device, pdn, _lib.lib(), rtol=1e-4, want_families=(2,))


from tests.conftest import device_variants  # noqa: E402
device_variants(globals(), check_batch_gate_logic)
