"""Checkpoint I/O of the Llama workload (reference llm/llama/io.py:8-57): HF-keyed .npz -> model
(projection matrices transposed, lm_head.bias untouched), finetuned-parameter round trip.  Runs on
the NumPy device and on the emulated HIP device (host logic + upload/download paths)."""
import numpy as np
import pytest

import pydynet_amd as pdn
from pydynet_amd.core.tensor import Graph
from pydynet_amd.llm.llama import Llama
from pydynet_amd.llm import io as lio

CFG = dict(V=64, D=48, H=2, F=96, L=16, B=2, layers=2)


def _model(dev):
    Graph.clear()
    np.random.seed(3)
    m = Llama(CFG["V"], CFG["D"], CFG["H"], CFG["F"], 32, CFG["B"], CFG["layers"], np.float32)
    m.tok_embedding.weight.data[...] = np.zeros((CFG["V"], CFG["D"]), np.float32)
    return m.to(dev)


def _host(a):
    return a if isinstance(a, np.ndarray) else a.get()


def _hf_checkpoint(path):
    rng = np.random.default_rng(5)
    D, F, V = CFG["D"], CFG["F"], CFG["V"]
    w = {"model.embed_tokens.weight": rng.standard_normal((V, D), dtype=np.float32),
         "lm_head.weight": rng.standard_normal((V, D), dtype=np.float32),          # (out, in)
         "model.norm.weight": rng.standard_normal(D, dtype=np.float32)}
    for i in range(CFG["layers"]):
        p = f"model.layers.{i}."
        for k in "qkvo":
            w[p + f"self_attn.{k}_proj.weight"] = rng.standard_normal((D, D), dtype=np.float32)
        w[p + "mlp.up_proj.weight"] = rng.standard_normal((F, D), dtype=np.float32)
        w[p + "mlp.gate_proj.weight"] = rng.standard_normal((F, D), dtype=np.float32)
        w[p + "mlp.down_proj.weight"] = rng.standard_normal((D, F), dtype=np.float32)
        w[p + "input_layernorm.weight"] = rng.standard_normal(D, dtype=np.float32)
        w[p + "post_attention_layernorm.weight"] = rng.standard_normal(D, dtype=np.float32)
    np.savez(path, **w)
    return w


@pytest.mark.parametrize("variant", ["cpu", "emulated"])
def test_load_hf_checkpoint_and_finetuned_round_trip(variant, tmp_path, request):
    if variant == "emulated":
        request.getfixturevalue("emulated_hip")
        dev = "hip:0"
    else:
        dev = "cpu"
    m = _model(dev)
    bias_before = _host(m.lm_head.bias.data).copy()
    w = _hf_checkpoint(tmp_path / "hf.npz")
    assert lio.load_model(m, str(tmp_path / "hf.npz")) is m
    P = m._parameters
    assert np.array_equal(_host(P["tok_embedding.weight"].data), w["model.embed_tokens.weight"])
    assert np.array_equal(_host(P["lm_head.weight"].data), w["lm_head.weight"].T)
    assert np.array_equal(_host(P["lm_head.bias"].data), bias_before)            # never in the file
    assert np.array_equal(_host(P["norm.weight"].data), w["model.norm.weight"])
    for i in range(CFG["layers"]):
        assert np.array_equal(_host(P[f"layers.{i}.attention.K.weight"].data),
                              w[f"model.layers.{i}.self_attn.k_proj.weight"].T)
        assert np.array_equal(_host(P[f"layers.{i}.ffn.down.weight"].data),
                              w[f"model.layers.{i}.mlp.down_proj.weight"].T)
        assert np.array_equal(_host(P[f"layers.{i}.post_attn_norm.weight"].data),
                              w[f"model.layers.{i}.post_attention_layernorm.weight"])
    # every mapped name exists and the map covers all weights a HF file holds
    assert set(lio.hf_key_map(CFG["layers"])) <= set(P)
    assert {k for k, _ in lio.hf_key_map(CFG["layers"]).values()} == set(w)

    # finetune only lm_head, save, perturb, load back
    m.set_trainable_parameters(("lm_head",))
    lio.save_finetuned_parameters(m, str(tmp_path / "ft.npz"))
    saved = np.load(tmp_path / "ft.npz")
    assert sorted(saved.files) == ["lm_head.bias", "lm_head.weight"]
    ref_w = _host(P["lm_head.weight"].data).copy()
    P["lm_head.weight"].data[...] = np.zeros_like(ref_w)
    lio.load_finetuned_parameters(m, str(tmp_path / "ft.npz"))
    assert np.array_equal(_host(P["lm_head.weight"].data), ref_w)

    bad = dict(w); bad["model.norm.weight"] = np.zeros(CFG["D"] + 1, np.float32)
    np.savez(tmp_path / "bad.npz", **bad)
    with pytest.raises(ValueError):
        lio.load_model(m, str(tmp_path / "bad.npz"))
