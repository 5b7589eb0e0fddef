"""Checkpoint I/O for the Llama workload (reference: llm/llama/io.py:8-57).

`load_model` reads a Hugging-Face-keyed `.npz` (the `stories15M.model.npz` format): projection
matrices are stored (out, in) there and (in, out) here (nn/modules/linear.py:26-27), so they are
transposed on the way in; norm weights and the embedding table are copied as they are; `lm_head.bias`
is never present in such a file and keeps its initial value (io.py:12-13).
`save_finetuned_parameters` / `load_finetuned_parameters` round-trip the trainable subset under our
own parameter names (`layers.{i}.attention.Q.weight`, ...).  Works for parameters on any device:
`param.data[...] = host_array` uploads, `param.numpy()` downloads.
"""
from __future__ import annotations

import numpy as np

from ..autograd import no_grad

# our sub-module path inside a block -> (HF key suffix, stored transposed?)
_BLOCK_KEYS = (
    ("attention.Q.weight", "self_attn.q_proj.weight", True),
    ("attention.K.weight", "self_attn.k_proj.weight", True),
    ("attention.V.weight", "self_attn.v_proj.weight", True),
    ("attention.O.weight", "self_attn.o_proj.weight", True),
    ("ffn.up.weight", "mlp.up_proj.weight", True),
    ("ffn.gate.weight", "mlp.gate_proj.weight", True),
    ("ffn.down.weight", "mlp.down_proj.weight", True),
    ("input_norm.weight", "input_layernorm.weight", False),
    ("post_attn_norm.weight", "post_attention_layernorm.weight", False),
)


def hf_key_map(n_layers: int):
    """{our parameter name: (HF key, transposed)} for an `n_layers` model."""
    table = {"tok_embedding.weight": ("model.embed_tokens.weight", False),
             "lm_head.weight": ("lm_head.weight", True),
             "norm.weight": ("model.norm.weight", False)}
    for i in range(n_layers):
        for ours, theirs, tr in _BLOCK_KEYS:
            table[f"layers.{i}.{ours}"] = (f"model.layers.{i}.{theirs}", tr)
    return table


def _assign(param, value):
    value = np.asarray(value)
    if tuple(value.shape) != tuple(param.shape):
        raise ValueError(f"checkpoint tensor has shape {value.shape}, parameter expects {tuple(param.shape)}")
    param.data[...] = np.ascontiguousarray(value, dtype=param.dtype)


@no_grad()
def load_model(llama, model_path: str):
    weights = np.load(model_path)
    params = llama._parameters
    for name, (key, transposed) in hf_key_map(llama.n_layers).items():
        w = weights[key]
        _assign(params[name], w.T if transposed else w)
    return llama


@no_grad()
def save_finetuned_parameters(model, output_path: str):
    np.savez(output_path, **{name: p.numpy() for name, p in model._parameters.items() if p.requires_grad})


@no_grad()
def load_finetuned_parameters(model, finetuned_path: str):
    weights = np.load(finetuned_path)
    for name, p in model._parameters.items():
        if name in weights:
            _assign(p, weights[name])
    return model
