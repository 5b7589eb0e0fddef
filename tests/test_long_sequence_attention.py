"""Training attention beyond 256 positions and at head dim 64 on the resident kernels (csrc/attention.hip: K / V --
or Q / dO -- of a head pass through LDS in 256-row chunks, the forward carries (m, l, O) across chunks with one online
rescale).  The reference's model allows max_seq_len 1024 (llm/llama/finetune.py:44, model.py:176-181).  A small Llama
is trained one step on the "cpu" device -- the NumPy composition pinned to the reference by tests/test_llama_golden.py
-- and on the HIP device; loss and every gradient must agree to the north-star tolerance, and the step must have gone
through the fused qkv_attention node on the RESIDENT kernels (not the streaming ones).  Runs on the real MI355X
(-m gpu) and on the emulated C ABI."""
import numpy as np

from tests.conftest import device_variants


def _step(dev, V, D, H, F_, L, B, seed):
    import pydynet_amd as pdn  # noqa: F401
    from pydynet_amd.llm.llama import Llama
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    np.random.seed(seed)
    m = Llama(V, D, H, F_, L, B, 1, np.float32)
    m.tok_embedding.weight.data[...] = (0.05 * np.random.randn(V, D)).astype(np.float32)
    m = m.to(dev)
    rng = np.random.default_rng(seed)
    ids, tgt = rng.integers(0, V, (B, L)), rng.integers(0, V, (B, L))
    m.train(True)
    loss = m.loss(ids, tgt)
    loss.backward()
    return float(loss.item()), {n: p.grad.get() if hasattr(p.grad, "get") else np.array(p.grad)
                                for n, p in m.named_parameters()}


def _check(dev, V, D, H, F_, L, B, seed):
    from pydynet_amd.core import fused
    kinds = []
    orig = fused.qkv_attention.forward_

    def spy(node, *a):
        out = orig(node, *a)
        kinds.append(fused.qkv_attention._resident(L, D // H))
        return out
    fused.qkv_attention.forward_ = spy
    try:
        loss, grads = _step(dev, V, D, H, F_, L, B, seed)
    finally:
        fused.qkv_attention.forward_ = orig
    assert kinds and all(kinds), "the step did not take the fused qkv_attention node on the resident kernels"
    ref_loss, ref = _step("cpu", V, D, H, F_, L, B, seed)
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss), (loss, ref_loss)
    for n, g in grads.items():
        scale = max(float(np.abs(ref[n]).max()), 1e-30)
        err = float(np.abs(g.astype(np.float64) - ref[n]).max())
        assert err <= 1e-4 * scale + 1e-7, (n, err, scale)


def check_llama_step_seq512_hd48(dev):
    _check(dev, 64, 96, 2, 128, 512, 2, 5)          # two 256-key chunks


def check_llama_step_seq352_hd64(dev):
    _check(dev, 64, 128, 2, 160, 352, 1, 6)         # head dim 64, ragged second chunk (11 tiles)


for _f in (check_llama_step_seq512_hd48, check_llama_step_seq352_hd64):
    device_variants(globals(), _f)
