"""Training attention beyond 256 positions and at head dim 64 on the resident kernels (csrc/attention.hip: K / V --
or Q / dO -- of a head pass through LDS in 256-row chunks, the forward carries (m, l, O) across chunks with one online
rescale).  The reference's model allows max_seq_len 1024 (llm/llama/finetune.py:44, model.py:176-181).

Pinned to the REAL reference (round 4): `tests/golden/long_attention.npz` holds the loss and every gradient of one
training step of a one-layer Llama at L = 512 / hd 48 (two 256-key chunks) and L = 352 / hd 64 (ragged second chunk),
produced by `tools/gen_golden_r2.py long_attention` importing /root/reference (llm/llama/model.py:23-44, 95-121,
226-252).  The same seeded model is stepped here on the "cpu" device, on the emulated C ABI and (-m gpu) on the real
MI355X; on the HIP devices the step must have gone through the fused qkv_attention node on the RESIDENT kernels (not
the streaming ones)."""
import os

import numpy as np

from tests.conftest import device_variants

G = os.path.join(os.path.dirname(__file__), "golden")
CASES = {  # tag: V, D, H, F, L, B, seed -- tools/gen_golden_r2.py LONG_CASES
    "seq512_hd48": (64, 96, 2, 128, 512, 2, 5),
    "seq352_hd64": (64, 128, 2, 160, 352, 1, 6),
}


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


def _check(dev, tag):
    from pydynet_amd.core import fused
    V, D, H, F_, L, B, seed = CASES[tag]
    ref = np.load(os.path.join(G, "long_attention.npz"))
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
    if dev != "cpu":
        assert kinds and all(kinds), "the step did not take the fused qkv_attention node on the resident kernels"
    ref_loss = float(ref[f"{tag}/loss"])
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss), (loss, ref_loss)
    names = [k[len(tag) + 6:] for k in ref.files if k.startswith(tag + "/grad/")]
    assert sorted(names) == sorted(grads), (sorted(names), sorted(grads))
    for n in names:
        r = ref[f"{tag}/grad/{n}"]
        scale = max(float(np.abs(r).max()), 1e-30)
        err = float(np.abs(grads[n].astype(np.float64) - r).max())
        assert err <= 1e-4 * scale + 1e-7, (n, err, scale)


def check_llama_step_seq512_hd48(dev):
    _check(dev, "seq512_hd48")          # two 256-key chunks


def check_llama_step_seq352_hd64(dev):
    _check(dev, "seq352_hd64")          # head dim 64, ragged second chunk (11 tiles)


def test_llama_step_seq512_hd48_cpu():
    _check("cpu", "seq512_hd48")


def test_llama_step_seq352_hd64_cpu():
    _check("cpu", "seq352_hd64")


def check_llama_step_seq512_hd48_prerotated(dev):
    """The same reference step (`long_attention.npz`, width 96: no RoPE store in the projection) with q | k rotated in
    place in the packed projection and the attention as 256-row block pairs on the PERSISTENT kernels (round 5:
    core/fused/attn.py `prerotated`, csrc/attention_blocks.hip).  The node's row threshold is lowered for the fixture's
    1024 tokens; the library's counters say which kernels ran."""
    import ctypes
    from pydynet_amd.core import fused
    from pydynet_amd import _lib
    saved, fused.qkv_attention.rope_min_rows = fused.qkv_attention.rope_min_rows, 32
    buf = (ctypes.c_int64 * 21)()
    try:
        _lib.lib().call("pdn_kernel_counters", buf, 21, 1)
        _check(dev, "seq512_hd48")
        _lib.lib().call("pdn_kernel_counters", buf, 21, 1)
    finally:
        fused.qkv_attention.rope_min_rows = saved
    assert buf[7] >= 1 and buf[8] >= 1 and buf[9] == 0 and buf[10] == 0, ("persistent / resident attention launches", list(buf[7:11]))


for _f in (check_llama_step_seq512_hd48, check_llama_step_seq352_hd64, check_llama_step_seq512_hd48_prerotated):
    device_variants(globals(), _f)
