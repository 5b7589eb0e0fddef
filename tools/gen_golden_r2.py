"""Round-2 golden vectors, produced by running the REAL reference (imported from /root/reference).

Same rules as tools/gen_golden.py: runs only in the build container, stores inputs and the
reference's outputs / gradients under tests/golden/ keyed by our own names, nothing of the
reference's source.  Fixtures written here:

    fused_llama.npz   hd = 48, L = 256 Llama (the shape class bench.py times): logits, 3 losses,
                      first-step gradients, final parameters  -> pins qkv_attention + RoPE-in-load +
                      multi-tile causal attention to llm/llama/model.py:23-44,95-121,226-252
    generate.npz      the reference's KV-cache `generate` (model.py:105-110,254-269) on a tiny
                      hd = 48 model: token ids and the logits of every step, batch 1 and batch 2
    llama_io.npz      the reference's `load_model` on a synthetic HF-keyed npz and its
                      `save_finetuned_parameters` output (llm/llama/io.py:8-57)
    clip_blocks.npz   llm/clip/model.py:35-80,83-113: biased MHA (hd = 64, with and without the
                      causal mask), last-axis LayerNorm, quick-GELU MLP, one Transformer block
    long_attention.npz   one training step (loss + all gradients) at L = 512 / hd 48 and L = 352 / hd 64: the chunked
                      resident attention kernels against llm/llama/model.py:95-121 (finetune.py:44 allows 1024)
    generate_full.json   64 greedy tokens of the FULL-width model (V 32000, D 288, 6 layers): ids + four scalars of
                      every step's logits row (model.py:254-269)
    wide_llama.npz    one training step of a one-layer Llama of width 512 / head dim 64 / ffn 1376 over 4096 tokens: loss, norm
                      and a strided sample of every gradient -> pins the SwiGLU epilogues of the tiled kernel (round 5)
    long288_llama.npz the same for width 288 / head dim 48 at 512 positions (4096 tokens) -> pins RoPE in the projection's
                      store + the attention as 256-row block pairs on the persistent kernels (round 5)
    masked_attention.npz   the attention chain of examples/pydynet/transformer.py:84-101 (matmul, / sqrt(hd), + (B, 1, 1, L)
                      padding mask with -inf, softmax, matmul) on the reference's own operators at head dim 48 / 64 and
                      whole 32-row tiles, non-causal: output and the three input gradients
    ops_r2.npz        split / vsplit / hsplit / dsplit (function.py:14-166) incl. gradients,
                      nll_loss (functional.py:353-361), float16 operator cases
                      (tests/test_tensor_basic.py:16,80-81)

    python tools/gen_golden_r2.py
"""
import os
import sys
import tempfile
import warnings

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

import pydynet as pdn                      # noqa: E402
import pydynet.nn as nn                    # noqa: E402
import pydynet.nn.functional as F          # noqa: E402
from pydynet.optim import Adam             # noqa: E402
from pydynet.core.tensor import Graph      # noqa: E402


def fresh():
    Graph.node_list.clear(); Graph.size = 0
    pdn.autograd.set_grad_enabled(True)


FUSED_CFG = dict(V=128, D=96, H=2, F=128, L=256, B=2, layers=2, seed=4321, lr=1e-3, steps=3, max_seq=256)


def gen_fused_llama():
    from llm.llama.model import Llama
    c = FUSED_CFG
    fresh()
    np.random.seed(c["seed"])
    m = Llama(c["V"], c["D"], c["H"], c["F"], c["max_seq"], c["B"], c["layers"], np.float32)
    m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(c["V"], c["D"])).astype(np.float32)
    # duplicate-free token columns are impossible at L = 256 > V: the scatter-ASSIGN embedding
    # gradient (last occurrence wins) is therefore part of what this fixture pins
    ids = np.random.randint(0, c["V"], (c["B"], c["L"]))
    tgt = np.random.randint(0, c["V"], (c["B"], c["L"]))
    names = [n for n, p in m._parameters.items() if p.requires_grad]
    d = {"ids": ids, "tgt": tgt}
    for n in names:
        d["init/" + n] = m._parameters[n].data.copy()
    m.train(True)
    with pdn.no_grad():
        pass
    logits = m.forward_logits(ids)
    d["logits0"] = logits.data.copy()
    fresh()
    opt = Adam(m.parameters(), lr=c["lr"])
    losses = []
    for s in range(c["steps"]):
        losses.append(m.finetune_step(ids, tgt, opt))
        if s == 0:
            for n in names:
                d["grad1/" + n] = m._parameters[n].grad.copy()
    for n in names:
        d["final/" + n] = m._parameters[n].data.copy()
    d["losses"] = np.array(losses, np.float64)
    np.savez_compressed(os.path.join(OUT, "fused_llama.npz"), **d)
    print("fused llama losses", losses)


def gen_generate():
    from llm.llama.model import Llama
    d = {}
    # p512: a 512-token prompt (prefill through multi-tile causal attention with RoPE positions up to 511,
    # the reference's RoPE table / KV cache go to 1024: llm/llama/model.py:86-93,176-181), then 3 decode steps
    for tag, B, Lp, total, max_seq in (("b1", 1, 5, 17, 64), ("b2", 2, 7, 15, 64), ("long", 1, 40, 48, 64),
                                       ("p512", 1, 512, 515, 520)):
        fresh()
        np.random.seed(99)
        V, D, H, Ff, layers = 96, 96, 2, 128, 2
        m = Llama(V, D, H, Ff, max_seq, B, layers, np.float32)
        m.tok_embedding.weight.data[...] = (0.5 * np.random.randn(V, D)).astype(np.float32)
        # sharpen the head so greedy argmax margins sit far above fp32 reassociation noise
        m.lm_head.weight.data[...] = (0.5 * np.random.randn(D, V)).astype(np.float32)
        prompt = np.random.randint(0, V, (B, Lp))
        if tag == "b1":
            for n, p in m._parameters.items():
                if p.requires_grad:
                    d["init/" + n] = p.data.copy()
        m.eval()
        logits_log = []
        orig_forward = m.forward

        def rec(input_ids, start_pos, _f=orig_forward, _log=logits_log):
            out = _f(input_ids, start_pos)
            _log.append(out.data.copy())
            return out
        m.forward = rec
        with pdn.no_grad():
            toks = [t.data.copy() for t in m.generate(prompt, total)]
        d[f"{tag}/prompt"] = prompt
        d[f"{tag}/total"] = np.array(total)
        d[f"{tag}/max_seq"] = np.array(max_seq)
        d[f"{tag}/tokens"] = np.concatenate(toks, axis=1)               # (B, total - Lp)
        d[f"{tag}/logits"] = np.concatenate(logits_log, axis=1)         # (B, total - Lp, V)
        top2 = np.sort(d[f"{tag}/logits"], axis=-1)[..., -2:]
        print("generate", tag, d[f"{tag}/tokens"][0][:8], "min argmax margin", float((top2[..., 1] - top2[..., 0]).min()))
    d["cfg"] = np.array([96, 96, 2, 128, 2, 64])
    np.savez_compressed(os.path.join(OUT, "generate.npz"), **d)
    pdn.autograd.set_grad_enabled(True)


def gen_llama_io():
    from llm.llama.model import Llama
    from llm.llama.io import load_model, save_finetuned_parameters, load_finetuned_parameters
    fresh()
    V, D, H, Ff, layers = 64, 48, 2, 96, 2
    rng = np.random.default_rng(5)
    hf = {"model.embed_tokens.weight": rng.standard_normal((V, D)).astype(np.float32),
          "lm_head.weight": rng.standard_normal((V, D)).astype(np.float32),
          "model.norm.weight": rng.standard_normal((D,)).astype(np.float32)}
    for i in range(layers):
        pre = f"model.layers.{i}."
        for k in ("q_proj", "k_proj", "v_proj", "o_proj"):
            hf[pre + f"self_attn.{k}.weight"] = rng.standard_normal((D, D)).astype(np.float32)
        hf[pre + "mlp.up_proj.weight"] = rng.standard_normal((Ff, D)).astype(np.float32)
        hf[pre + "mlp.gate_proj.weight"] = rng.standard_normal((Ff, D)).astype(np.float32)
        hf[pre + "mlp.down_proj.weight"] = rng.standard_normal((D, Ff)).astype(np.float32)
        hf[pre + "input_layernorm.weight"] = rng.standard_normal((D,)).astype(np.float32)
        hf[pre + "post_attention_layernorm.weight"] = rng.standard_normal((D,)).astype(np.float32)
    d = {"hf/" + k: v for k, v in hf.items()}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "hf.npz")
        np.savez(path, **hf)
        np.random.seed(3)
        m = load_model(Llama(V, D, H, Ff, 32, 1, layers, np.float32), path)
        for n, p in m._parameters.items():
            if p.requires_grad and n != "lm_head.bias":           # bias is never loaded (random init)
                d["loaded/" + n] = p.data.copy()
        # first-token logits of the loaded model (end-to-end check of the transposes)
        m.eval()
        with pdn.no_grad():
            d["prompt"] = np.array([[1, 7, 3, 9]])
            m.lm_head.bias.data[...] = 0
            d["logits"] = m(d["prompt"], 0).data.copy()
        pdn.autograd.set_grad_enabled(True)
        m.set_trainable_parameters(("lm_head", "layers.1.ffn"))
        out = os.path.join(tmp, "ft.npz")
        save_finetuned_parameters(m, out)
        saved = np.load(out)
        d["saved_keys"] = np.array(sorted(saved.files))
        for k in saved.files:
            d["saved/" + k] = saved[k]
        # load_finetuned_parameters assigns by name
        np.random.seed(4)
        m2 = Llama(V, D, H, Ff, 32, 1, layers, np.float32)
        before = m2.layers[0].ffn.up.weight.data.copy()
        load_finetuned_parameters(m2, out)
        assert np.array_equal(m2.lm_head.weight.data, saved["lm_head.weight"])
        assert np.array_equal(m2.layers[0].ffn.up.weight.data, before)
    np.savez_compressed(os.path.join(OUT, "llama_io.npz"), **d)
    print("llama_io", len(d), "arrays")


def gen_clip_blocks():
    from llm.clip.model import MultiHeadAttention, CLIPLayerNorm, MLP, Transformer, build_attention_mask
    d = {}
    B, L, D, H, M = 2, 40, 128, 2, 256                       # hd = 64 as in CLIP ViT-B/32
    rng = np.random.default_rng(11)
    x_np = rng.standard_normal((B, L, D)).astype(np.float32)
    w_np = rng.standard_normal((B, L, D)).astype(np.float32)         # loss weights
    d["x"], d["w"] = x_np, w_np

    def run(tag, module, call):
        fresh()
        x = pdn.Tensor(x_np, dtype=np.float32, requires_grad=True)
        y = call(module, x)
        d[f"{tag}/y"] = y.data.copy()
        (y * pdn.Tensor(w_np[tuple(slice(0, s) for s in y.shape)], dtype=np.float32)).sum().backward()
        d[f"{tag}/dx"] = x.grad.copy()
        for n, p in module._parameters.items():
            if p.requires_grad:
                d[f"{tag}/p/{n}"] = p.data.copy()
                d[f"{tag}/g/{n}"] = p.grad.copy()

    def randomise(module, seed):
        r = np.random.default_rng(seed)
        for n, p in module._parameters.items():
            if p.requires_grad:
                p.data[...] = (r.standard_normal(p.shape) * (0.1 if p.ndim > 1 else 0.5) + (1.0 if "scale" in n else 0.0)).astype(np.float32)

    np.random.seed(21)
    mha = MultiHeadAttention(D, H); randomise(mha, 1)
    run("mha_nomask", mha, lambda m, x: m(x, None))
    mha = MultiHeadAttention(D, H); randomise(mha, 1)
    mask = build_attention_mask(L)
    run("mha_causal", mha, lambda m, x: m(x, mask))
    ln = CLIPLayerNorm((D,), eps=1e-5, dtype=np.float32); randomise(ln, 2)
    run("layernorm", ln, lambda m, x: m(x))
    mlp = MLP(D, M); randomise(mlp, 3)
    run("mlp", mlp, lambda m, x: m(x))
    blk = Transformer(D, H, M); randomise(blk, 4)
    run("block", blk, lambda m, x: m(x, mask))
    np.savez_compressed(os.path.join(OUT, "clip_blocks.npz"), **d)
    print("clip_blocks", len(d), "arrays")


def gen_ops_r2():
    d = {}
    rng = np.random.default_rng(123)
    x_np = rng.standard_normal((4, 6, 8, 2)).astype(np.float32)
    d["split/x"] = x_np
    cases = [("split_a0", lambda x: pdn.split(x, 2, axis=0)), ("split_a1", lambda x: pdn.split(x, 3, axis=1)),
             ("split_a2", lambda x: pdn.split(x, 4, axis=2)), ("split_a3", lambda x: pdn.split(x, 2, axis=3)),
             ("split_idx", lambda x: pdn.split(x, (1, 4), axis=1)), ("vsplit", lambda x: pdn.vsplit(x, 2)),
             ("hsplit", lambda x: pdn.hsplit(x, (2, 3))), ("dsplit", lambda x: pdn.dsplit(x, 2))]
    for tag, fn in cases:
        fresh()
        x = pdn.Tensor(x_np, dtype=np.float32, requires_grad=True)
        parts = fn(x)
        d[f"{tag}/n"] = np.array(len(parts))
        loss = None
        for i, p in enumerate(parts):
            d[f"{tag}/{i}"] = p.data.copy()
            term = (p * float(i + 1)).sum()
            loss = term if loss is None else loss + term
        loss.backward()
        d[f"{tag}/dx"] = x.grad.copy()
    # nll_loss (functional.py:353-361): mean / sum of -y_pred * y_true, used after log_softmax
    for red in ("mean", "sum"):
        fresh()
        logits = pdn.Tensor(rng.standard_normal((6, 5)).astype(np.float32), dtype=np.float32, requires_grad=True)
        onehot = np.eye(5, dtype=np.float32)[rng.integers(0, 5, 6)]
        lp = F.log_softmax(logits, axis=1, keepdims=True)
        loss = F.nll_loss(lp, pdn.Tensor(onehot, dtype=np.float32), reduction=red)
        loss.backward()
        d[f"nll_{red}/logits"], d[f"nll_{red}/onehot"] = logits.data.copy(), onehot
        d[f"nll_{red}/loss"], d[f"nll_{red}/dlogits"] = np.asarray(loss.data).copy(), logits.grad.copy()
    # float16 operator cases: the reference's tests draw f16 operands (tests/test_tensor_basic.py:16)
    a = rng.standard_normal((3, 1, 5)).astype(np.float16)
    b = (rng.standard_normal((4, 5)) + 2.5).astype(np.float16)
    c = rng.standard_normal((4, 5)).astype(np.float32)
    d["f16/a"], d["f16/b"], d["f16/c"] = a, b, c
    with np.errstate(all="ignore"):
        for n in ("add", "sub", "mul", "div", "maximum", "minimum"):
            fresh()
            d[f"f16/{n}_hh"] = getattr(pdn, n)(pdn.Tensor(a), pdn.Tensor(b)).data
            d[f"f16/{n}_hs"] = getattr(pdn, n)(pdn.Tensor(a), pdn.Tensor(c)).data
        fresh()
        d["f16/pow_hh"] = pdn.pow(pdn.Tensor(np.abs(a) + np.float16(0.5)), pdn.Tensor(b)).data
        for n in ("exp", "log", "abs", "sign"):
            fresh()
            src = np.abs(b) if n == "log" else a
            d[f"f16/{n}"] = getattr(pdn, n)(pdn.Tensor(src)).data
        for n, kw in (("sum", dict(axis=-1)), ("mean", dict(axis=0)), ("max", dict(axis=(0, 2))), ("min", dict())):
            fresh()
            d[f"f16/r_{n}"] = np.asarray(getattr(pdn, n)(pdn.Tensor(a), **kw).data)
        fresh()
        d["f16/matmul"] = pdn.matmul(pdn.Tensor(b), pdn.Tensor(b.T.copy())).data
        # a small f16 training-style graph with gradients
        fresh()
        xa = pdn.Tensor(a, dtype=np.float16, requires_grad=True)
        xb = pdn.Tensor(b, dtype=np.float16, requires_grad=True)
        y = (xa * xb + pdn.exp(xa)).sum()
        y.backward()
        d["f16/g_out"], d["f16/g_a"], d["f16/g_b"] = np.asarray(y.data), xa.grad.copy(), xb.grad.copy()
    np.savez_compressed(os.path.join(OUT, "ops_r2.npz"), **d)
    print("ops_r2", len(d), "arrays")


LONG_CASES = {  # tag: V, D, H, F, L, B, seed  (tests/test_long_sequence_attention.py builds the same models)
    "seq512_hd48": (64, 96, 2, 128, 512, 2, 5),
    "seq352_hd64": (64, 128, 2, 160, 352, 1, 6),
}


def gen_long_attention():
    """One training step (loss + every gradient) of a one-layer Llama beyond 256 positions / at head dim 64 on the REAL
    reference: pins the chunked resident attention kernels (round 3) to llm/llama/model.py:23-44, 95-121, 226-252
    (the reference's own max_seq_len is 1024, finetune.py:44)."""
    from llm.llama.model import Llama
    d = {}
    for tag, (V, D, H, Ff, L, B, seed) in LONG_CASES.items():
        fresh()
        np.random.seed(seed)
        m = Llama(V, D, H, Ff, L, B, 1, np.float32)
        m.tok_embedding.weight.data[...] = (0.05 * np.random.randn(V, D)).astype(np.float32)
        rng = np.random.default_rng(seed)
        ids, tgt = rng.integers(0, V, (B, L)), rng.integers(0, V, (B, L))
        m.train(True)
        logits = m.forward_logits(ids)
        loss = F.cross_entropy_loss(logits.reshape(B * L, V), pdn.Tensor(tgt.reshape(-1), dtype=np.int64))
        loss.backward()
        d[f"{tag}/loss"] = np.array(float(loss.item()))
        for n, p in m._parameters.items():
            if p.requires_grad:
                d[f"{tag}/grad/{n}"] = np.asarray(p.grad, np.float32).copy()
        print("long attention", tag, float(loss.item()), len([k for k in d if k.startswith(tag + "/grad/")]), "gradients")
    np.savez_compressed(os.path.join(OUT, "long_attention.npz"), **d)


WIDE_CASE = dict(V=256, D=512, H=8, F=1376, L=256, B=16, seed=11)    # tests/test_wide_llama.py builds the same model


def gen_wide_llama():
    """One training step of a one-layer Llama of width 512 (head dim 64, ffn 1376: not a multiple of the 128 / 256-column
    tiles) over 4096 tokens on the REAL reference: loss, the norm of every gradient and a strided sample of its entries.
    Pins the SwiGLU epilogues of the tiled kernel (round 5, csrc/gemm.hip SWI) -- the model widths the row-resident
    kernels do not take -- to llm/llama/model.py:47-58, 153-197."""
    from llm.llama.model import Llama
    c = WIDE_CASE
    fresh()
    np.random.seed(c["seed"])
    m = Llama(c["V"], c["D"], c["H"], c["F"], c["L"], c["B"], 1, np.float32)
    m.tok_embedding.weight.data[...] = (0.05 * np.random.randn(c["V"], c["D"])).astype(np.float32)
    rng = np.random.default_rng(c["seed"])
    ids, tgt = rng.integers(0, c["V"], (c["B"], c["L"])), rng.integers(0, c["V"], (c["B"], c["L"]))
    m.train(True)
    logits = m.forward_logits(ids)
    loss = F.cross_entropy_loss(logits.reshape(c["B"] * c["L"], c["V"]), pdn.Tensor(tgt.reshape(-1), dtype=np.int64))
    loss.backward()
    d = {"loss": np.array(float(loss.item()))}
    for n, p in m._parameters.items():
        if p.requires_grad:
            g = np.asarray(p.grad, np.float64).reshape(-1)
            d["gnorm/" + n] = np.array(float(np.linalg.norm(g)))
            d["gmax/" + n] = np.array(float(np.abs(g).max()))
            d["gsample/" + n] = g[::61][:4096].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "wide_llama.npz"), **d)
    print("wide llama loss", float(loss.item()), len([k for k in d if k.startswith("gnorm/")]), "gradients")


LONG288_CASE = dict(V=64, D=288, H=6, F=768, L=512, B=8, seed=12)    # tests/test_wide_llama.py builds the same model


def gen_long288_llama():
    """One training step of a one-layer Llama of the BENCHMARKED width (288, head dim 48) at 512 positions over 4096 tokens
    on the REAL reference: loss, norm and a strided sample of every gradient.  At this width RoPE rides in the projection's
    store and the attention runs as 256-row block pairs on the persistent kernels (round 5, csrc/attention_blocks.hip):
    this fixture pins that path to llm/llama/model.py:23-44, 95-121."""
    from llm.llama.model import Llama
    c = LONG288_CASE
    fresh()
    np.random.seed(c["seed"])
    m = Llama(c["V"], c["D"], c["H"], c["F"], c["L"], c["B"], 1, np.float32)
    m.tok_embedding.weight.data[...] = (0.05 * np.random.randn(c["V"], c["D"])).astype(np.float32)
    rng = np.random.default_rng(c["seed"])
    ids, tgt = rng.integers(0, c["V"], (c["B"], c["L"])), rng.integers(0, c["V"], (c["B"], c["L"]))
    m.train(True)
    logits = m.forward_logits(ids)
    loss = F.cross_entropy_loss(logits.reshape(c["B"] * c["L"], c["V"]), pdn.Tensor(tgt.reshape(-1), dtype=np.int64))
    loss.backward()
    d = {"loss": np.array(float(loss.item()))}
    for n, p in m._parameters.items():
        if p.requires_grad:
            g = np.asarray(p.grad, np.float64).reshape(-1)
            d["gnorm/" + n] = np.array(float(np.linalg.norm(g)))
            d["gmax/" + n] = np.array(float(np.abs(g).max()))
            d["gsample/" + n] = g[::61][:4096].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "long288_llama.npz"), **d)
    print("long288 llama loss", float(loss.item()), len([k for k in d if k.startswith("gnorm/")]), "gradients")


GEN_FULL = dict(V=32000, D=288, H=6, F=768, layers=6, max_seq=128, seed=0, prompt_len=8, total=72)


def gen_generate_full():
    """The reference's greedy KV-cache `generate` (llm/llama/model.py:105-110, 254-269; the loop infer.py:46-63 times)
    at the FULL benchmark width: 64 new tokens; per step the token id and four scalars of the logits row (max, sum, l2
    norm, top-2 margin) -- scalars only, no arrays of the 32000-wide rows."""
    import json
    from llm.llama.model import Llama
    c = GEN_FULL
    fresh()
    np.random.seed(c["seed"])
    m = Llama(c["V"], c["D"], c["H"], c["F"], c["max_seq"], 1, c["layers"], np.float32)
    m.tok_embedding.weight.data[...] = (0.5 * np.random.randn(c["V"], c["D"])).astype(np.float32)
    m.lm_head.weight.data[...] = (0.5 * np.random.randn(c["D"], c["V"])).astype(np.float32)     # argmax margins >> fp32 noise
    prompt = np.random.randint(0, c["V"], (1, c["prompt_len"]))
    m.eval()
    rows = []
    orig_forward = m.forward

    def rec(input_ids, start_pos, _f=orig_forward):
        out = _f(input_ids, start_pos)
        r = out.data[0, -1].astype(np.float64)
        top2 = np.sort(r)[-2:]
        rows.append([float(r.max()), float(r.sum()), float(np.linalg.norm(r)), float(top2[1] - top2[0])])
        return out
    m.forward = rec
    with pdn.no_grad():
        toks = [int(t.data[0, 0]) for t in m.generate(prompt, c["total"])]
    pdn.autograd.set_grad_enabled(True)
    out = {"config": c, "prompt": prompt[0].tolist(), "tokens": toks, "logit_max": [r[0] for r in rows],
           "logit_sum": [r[1] for r in rows], "logit_l2": [r[2] for r in rows], "top2_margin": [r[3] for r in rows]}
    json.dump(out, open(os.path.join(OUT, "generate_full.json"), "w"), indent=1)
    print("generate full:", len(toks), "tokens", toks[:8], "min top-2 margin", min(out["top2_margin"]),
          "logit scale", max(out["logit_max"]))


MASKED_CASES = {"hd48": (2, 64, 2, 48, 11), "hd64": (2, 96, 1, 64, 12),       # B, L, H, hd, seed
                "hd128": (2, 44, 2, 128, 13)}      # the example's own head dim and length (dim 512 / 4 heads, L = 44: csrc/attention_hd128.hip)


def gen_masked_attention():
    """The Transformer example's attention chain with its padding mask on the REAL reference's operators
    (examples/pydynet/transformer.py:84-101; nn/functional.py:43-49 softmax): pins the key-bias form of the resident
    attention kernels (round 4)."""
    d = {}
    for tag, (B, L, H, hd, seed) in MASKED_CASES.items():
        fresh()
        rng = np.random.default_rng(seed)
        q, k, v, w = (rng.standard_normal((B, L, H, hd)).astype(np.float32) for _ in range(4))
        pad = np.zeros((B, 1, 1, L), np.float32)
        for b in range(B):
            pad[b, 0, 0, L - 3 - 7 * b:] = 1.0                         # construct_mask: 1 where the token is padding
        tq, tk, tv = (pdn.Tensor(a, dtype=np.float32, requires_grad=True) for a in (q, k, v))
        mask = pdn.Tensor(pad.copy(), dtype=np.float32)
        xq, xkT = tq.transpose(0, 2, 1, 3), tk.transpose(0, 2, 3, 1)
        att = xq @ xkT / hd ** .5
        mask[mask.eq(1)] = np.float32("-inf")                           # transformer.py:93
        att = att + mask
        att = F.softmax(att, axis=-1)
        out = (att @ tv.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)
        (out * pdn.Tensor(w, dtype=np.float32)).sum().backward()
        for n, a in (("q", q), ("k", k), ("v", v), ("w", w), ("pad", pad), ("out", out.data), ("dq", tq.grad), ("dk", tk.grad),
                     ("dv", tv.grad)):
            d[f"{tag}/{n}"] = np.asarray(a, np.float32).copy()
        print("masked attention", tag, float(np.abs(out.data).max()), float(np.abs(tk.grad).max()))
    np.savez_compressed(os.path.join(OUT, "masked_attention.npz"), **d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fused_llama", "generate", "llama_io", "clip_blocks", "ops_r2", "long_attention", "generate_full", "wide_llama", "long288_llama",
                             "masked_attention"]
    for w in which:
        globals()["gen_" + w]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
