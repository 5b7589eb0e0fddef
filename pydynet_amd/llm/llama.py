"""6-layer Llama3 -- the north-star workload (llm/llama/model.py of the reference), restated
on this package's fused nodes.

Same constructor signature, same registered parameter names (`layers.{i}.attention.Q.weight`,
`layers.{i}.ffn.gate.weight`, `lm_head.bias`, ...), same construction order (so a seeded NumPy
RNG yields the reference's initial weights), same `forward_logits / finetune_step / forward /
generate` entry points.  What differs is the number of tape nodes per block: 62 in the
reference, 8 here on the training path (rms_norm, qkv_attention [3 projections + RoPE + causal
attention], linear+residual, rms_norm, 2x linear, swiglu, linear+residual), each a handful of HIP
kernels / GEMMs per direction; the separate linear / rope / attention nodes remain for shapes the
fused attention kernels do not cover and for the KV-cache path, whose per-token step bypasses the
tape altogether (`_decode_step_hip`).
"""
import math
import os

import numpy as np

from .. import nn
from ..autograd import is_grad_enable
from ..core import Tensor, fused
from ..special import zeros


def compute_cos_sin_cache(head_dim: int, max_seq_len: int, base: int = 10000, dtype=None):
    """cos/sin(outer(arange(max_seq_len), base^(-2i/head_dim))) -> two (max_seq_len, head_dim/2) tensors."""
    inv_freq = 1.0 / (base ** (np.arange(0, head_dim, 2)[: head_dim // 2] / head_dim))
    freqs = np.outer(np.arange(max_seq_len), inv_freq).astype(dtype)
    return Tensor(np.cos(freqs)), Tensor(np.sin(freqs))


def apply_rotary_emb(xq, xk, freqs_cos, freqs_sin):
    """Rotate interleaved pairs (x[2i], x[2i+1]) of q and k by the position angle: one fused node each."""
    return fused.rope(xq, freqs_cos, freqs_sin), fused.rope(xk, freqs_cos, freqs_sin)


class FeedForward(nn.Module):
    def __init__(self, dim, up_dim, dtype=None):
        super().__init__()
        self.dim, self.up_dim = dim, up_dim
        self.up = nn.Linear(dim, up_dim, bias=False, dtype=dtype)
        self.gate = nn.Linear(dim, up_dim, bias=False, dtype=dtype)
        self.down = nn.Linear(up_dim, dim, bias=False, dtype=dtype)

    def move(self, device):
        super().move(device)
        if device.is_hip:
            self._pack_gate_up()
        return self

    def _pack_gate_up(self):
        """Re-home the gate / up weights in one (2, dim, ffn) buffer: each stays a contiguous Parameter,
        and being equally spaced lets both projections run as one batched GEMM."""
        from .. import hipnp as hp
        ws = [self.gate.weight, self.up.weight]
        st = hp.stacked_view([w.data for w in ws])
        if st is not None and st._strides[0] == ws[0].data.size:      # already adjacent, gate first
            return
        buf = hp.empty((2,) + tuple(ws[0].shape), ws[0].dtype)
        for i, w in enumerate(ws):
            buf[i] = w.data
            w.data = buf[i]

    def forward(self, x, residual=None):
        if fused.ffn_swiglu.applicable(x, self.gate.weight, self.up.weight, self.down.weight):
            # the whole block as one node: SwiGLU forward / backward ride in the projections' epilogues
            return fused.ffn_swiglu(x, self.gate.weight, self.up.weight, self.down.weight, residual)
        if fused.gate_up_swiglu.applicable(x, self.gate.weight, self.up.weight):
            h = fused.gate_up_swiglu(x, self.gate.weight, self.up.weight)
        else:
            h = fused.swiglu(self.gate(x), self.up(x))
        if residual is None:
            return self.down(h)
        return fused.linear(h, self.down.weight, None, residual)      # residual add in the GEMM epilogue


class Attention(nn.Module):
    def __init__(self, dim, n_heads, max_seq_len, max_batch_size=None, dtype=None):
        super().__init__()
        assert dim % n_heads == 0
        self.dim, self.n_heads, self.head_dim = dim, n_heads, dim // n_heads
        self.Q = nn.Linear(dim, dim, bias=False, dtype=dtype)
        self.K = nn.Linear(dim, dim, bias=False, dtype=dtype)
        self.V = nn.Linear(dim, dim, bias=False, dtype=dtype)
        self.O = nn.Linear(dim, dim, bias=False, dtype=dtype)
        self.max_seq_len = max_seq_len
        self.max_batch_size = max_batch_size if max_batch_size is not None else 1
        shape = (self.max_batch_size, max_seq_len, n_heads, self.head_dim)
        self.cache_k = nn.Parameter(zeros(shape, dtype=dtype), requires_grad=False)
        self.cache_v = nn.Parameter(zeros(shape, dtype=dtype), requires_grad=False)

    def move(self, device):
        super().move(device)
        if device.is_hip:
            self._pack_qkv()
        return self

    def _pack_qkv(self):
        """Re-home the Q/K/V weights in one (3, dim, dim) buffer: each stays a contiguous Parameter,
        and being equally spaced lets the three projections run as one batched GEMM."""
        from .. import hipnp as hp
        ws = [self.Q.weight, self.K.weight, self.V.weight]
        st = hp.stacked_view([w.data for w in ws])
        if st is not None and st._strides[0] == ws[0].data.size:      # already adjacent, in order
            return
        buf = hp.empty((3,) + tuple(ws[0].shape), ws[0].dtype)
        for i, w in enumerate(ws):
            buf[i] = w.data
            w.data = buf[i]

    def __call__(self, x, start_pos, mask, freqs_cos, freqs_sin, residual=None):
        B, L, _ = x.shape
        H, hd = self.n_heads, self.head_dim
        if (self._train and start_pos == 0 and mask is not None and is_grad_enable()
                and fused.qkv_attention.applicable(x, L, hd)):
            out = fused.qkv_attention(x, self.Q.weight, self.K.weight, self.V.weight, freqs_cos, freqs_sin, H)
            out = out.reshape(B, L, -1)
            if residual is None:
                return self.O(out)
            return fused.linear(out, self.O.weight, None, residual)
        xq = self.Q(x).reshape(B, L, H, hd)
        xk = self.K(x).reshape(B, L, H, hd)
        xv = self.V(x).reshape(B, L, H, hd)
        xq, xk = apply_rotary_emb(xq, xk, freqs_cos, freqs_sin)
        if not self._train:                       # KV cache: inference only (model.py:105-110)
            self.cache_k[:B, start_pos:start_pos + L] = xk
            self.cache_v[:B, start_pos:start_pos + L] = xv
            xk = self.cache_k[:B, :start_pos + L]
            xv = self.cache_v[:B, :start_pos + L]
        out = fused.attention(xq, xk, xv, causal=mask is not None, start_pos=start_pos)
        out = out.reshape(B, L, -1)
        if residual is None:
            return self.O(out)
        return fused.linear(out, self.O.weight, None, residual)


class TransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, ffn_dim, max_seq_len, max_batch_size=None, dtype=None):
        super().__init__()
        self.attention = Attention(dim, n_heads, max_seq_len, max_batch_size, dtype)
        self.ffn = FeedForward(dim, ffn_dim, dtype)
        self.input_norm = nn.RMSNorm(dim, dtype=dtype)
        self.post_attn_norm = nn.RMSNorm(dim, dtype=dtype)

    def forward(self, x, start_pos, mask, freqs_cos, freqs_sin):
        # z = x + attn(norm(x)); out = z + ffn(norm(z)) -- both adds ride in the GEMM epilogues
        z = self.attention(self.input_norm(x), start_pos, mask, freqs_cos, freqs_sin, residual=x)
        return self.ffn(self.post_attn_norm(z), residual=z)


class Llama(nn.Module):
    def __init__(self, vocab_size, embed_dim, n_heads, ffn_dim, max_seq_len, max_batch_size=None,
                 n_layers=6, dtype=None):
        super().__init__()
        self.vocab_size, self.embed_dim, self.n_heads, self.ffn_dim = vocab_size, embed_dim, n_heads, ffn_dim
        self.max_seq_len, self.max_batch_size, self.n_layers = max_seq_len, max_batch_size, n_layers
        self.tok_embedding = nn.Embedding(vocab_size, embed_dim, dtype=dtype)
        cos, sin = compute_cos_sin_cache(embed_dim // n_heads, max_seq_len, dtype=dtype)
        self.freqs_cos = nn.Parameter(cos, False)
        self.freqs_sin = nn.Parameter(sin, False)
        self.layers = nn.ModuleList([
            TransformerBlock(embed_dim, n_heads, ffn_dim, max_seq_len, max_batch_size, dtype)
            for _ in range(n_layers)])
        self.norm = nn.RMSNorm(embed_dim, dtype=dtype)
        self.lm_head = nn.Linear(embed_dim, vocab_size, dtype=dtype)      # bias=True, as the reference

    def _forward_hidden(self, input_ids, start_pos: int):
        L = input_ids.shape[-1]
        h = self.tok_embedding(input_ids)
        cos = self.freqs_cos[start_pos:start_pos + L]
        sin = self.freqs_sin[start_pos:start_pos + L]
        # the reference rebuilds an additive -inf mask on the host every call (model.py:199-203);
        # here causality is a flag of the fused attention node and nothing is uploaded.
        mask = True if L > 1 else None
        for layer in self.layers:
            h = layer(h, start_pos, mask, cos, sin)
        return self.norm(h)

    def forward_logits(self, input_ids, start_pos: int = 0):
        return self.lm_head(self._forward_hidden(input_ids, start_pos))

    def set_trainable_parameters(self, trainable_prefixes=("lm_head",)):
        trainable = frozen = 0
        for name, p in self._parameters.items():
            p.requires_grad = any(name.startswith(pre) for pre in trainable_prefixes)
            trainable, frozen = trainable + p.requires_grad, frozen + (not p.requires_grad)
        return trainable, frozen

    def loss(self, input_ids, target_ids, criterion=None, start_pos: int = 0):
        h = self._forward_hidden(input_ids, start_pos)
        if isinstance(target_ids, Tensor):
            targets = target_ids.reshape(-1)
        else:
            targets = Tensor(np.asarray(target_ids).reshape(-1), dtype=np.int64, device=h.device)
        head = self.lm_head
        bias = getattr(head, "bias", None)
        reduction = getattr(criterion, "reduction", "mean") if criterion is not None else "mean"
        if ((criterion is None or type(criterion) is nn.CrossEntropyLoss) and type(head) is nn.Linear
                and fused.linear_cross_entropy.applicable(h, head.weight, bias, targets, reduction)):
            # lm_head + cross entropy as one node: the (tokens, vocab) gradient of the logits is never written
            return fused.linear_cross_entropy(h, head.weight, bias, targets, reduction)
        logits = head(h)
        B, L, V = logits.shape
        return (criterion or nn.CrossEntropyLoss())(logits.reshape(B * L, V), targets)

    def finetune_step(self, input_ids, target_ids, optimizer, criterion=None, start_pos: int = 0):
        """zero_grad -> forward -> cross entropy -> backward -> optimizer step; returns the loss."""
        self.train(True)
        optimizer.zero_grad()
        loss = self.loss(input_ids, target_ids, criterion, start_pos)
        loss.backward()
        optimizer.step()
        return loss.item()

    def forward(self, input_ids, start_pos: int):
        return self.lm_head(self._forward_hidden(input_ids, start_pos)[:, [-1], :])

    fast_decode = True      # class switch: False keeps every decode step on the generic tape-node path

    def generate(self, input_ids, max_new_tokens: int):
        B, L = input_ids.shape
        next_id = None
        for i, pos in enumerate(range(L, max_new_tokens)):
            if i == 0:
                logits = self(input_ids, 0)                       # prompt pass: fills the KV caches
                next_id = logits[:, -1, :].argmax(-1, True)
            elif (Llama.fast_decode and next_id.device.is_hip and not self._train
                  and self.lm_head.weight.dtype == np.float32 and (self.embed_dim // self.n_heads) % 4 == 0):
                # (`more`: another token will be asked for -- the step after this one may be queued ahead)
                next_id = Tensor(self._decode_step_hip(next_id.data, pos, more=pos + 1 < max_new_tokens), dtype=np.int64,
                                 device=next_id.device, copy=False)
            else:
                next_id = self(next_id, pos)[:, -1, :].argmax(-1, True)
            yield next_id

    # -- decode fast path -----------------------------------------------------------
    graph_decode = True     # class switch: False issues the step's launches one by one instead of replaying a hipGraph
    decode_ahead = True     # class switch: False never queues the next step before the caller asked for it
    fused_decode = 2        # class switch: launches per layer = 2 (q|k|v inside the attention kernel), 1 -> 3, 0 / False -> 5

    def _decode_plan(self, B):
        """Buffers and weight views of the graph-replayable decode step (csrc/decode.hip), or None when the
        model's shapes / layout are outside what those kernels take (then the generic launches below run)."""
        from .. import hipnp as hp, _lib
        D, H, F, V = self.embed_dim, self.n_heads, self.ffn_dim, self.vocab_size
        st = getattr(self, "_decode_st", None)
        # a captured step (and the stacked weight views) hold the address of EVERY array the launches read: the key
        # covers them all -- rebinding `.data` of any parameter / cache re-plans -- and the switches that shape the plan
        ptrs = [self.lm_head.weight.data._ptr, self.tok_embedding.weight.data._ptr, self.norm.weight.data._ptr,
                self.freqs_cos.data._ptr, self.freqs_sin.data._ptr]
        bias = getattr(self.lm_head, "bias", None)
        ptrs.append(bias.data._ptr if bias is not None else 0)
        for layer in self.layers:
            a, f = layer.attention, layer.ffn
            ptrs += [t.data._ptr for t in (a.Q.weight, a.K.weight, a.V.weight, a.O.weight, a.cache_k, a.cache_v,
                                            f.gate.weight, f.up.weight, f.down.weight, layer.input_norm.weight,
                                            layer.post_attn_norm.weight)]
        key = (B, hp._state["device"], int(Llama.fused_decode or 0), os.environ.get("PDN_DECODE_SPLITS", ""),
               self.layers[0].attention.cache_k.shape[1], tuple(ptrs))      # (the addresses themselves: no hash to collide)
        if st is not None and st["key"] == key:
            return st if st["ok"] else None
        if st is not None:
            for g in st.get("graphs", {}).values():
                g.destroy()
        ok = (B <= 8 and B * max(D, F) <= 16384 and D % 4 == 0 and F % 4 == 0 and V % 4 == 0 and (D // H) % 4 == 0
              and self.layers[0].attention.cache_k.shape[1] * 4 <= 60 * 1024 and D // H <= 256)
        packs = []
        if ok:
            for layer in self.layers:
                a, f = layer.attention, layer.ffn
                qkv = hp.stacked_view([a.Q.weight.data, a.K.weight.data, a.V.weight.data])
                gu = hp.stacked_view([f.gate.weight.data, f.up.weight.data])
                mats = (a.O.weight.data, f.down.weight.data)
                if qkv is None or gu is None or not all(m.is_contiguous() for m in mats):    # (block strides may be < 0)
                    ok = False
                    break
                packs.append((qkv, gu))
            ok = ok and self.lm_head.weight.data.is_contiguous() and self.tok_embedding.weight.data.is_contiguous()
        st = {"B": B, "key": key, "ok": ok}
        if ok:
            nblk = _lib.lib().query("pdn_decode_gemv_blocks", V)
            # key ranges per head in the decode attention: one CU pulls ~11 B/clk, so long caches are cut up
            ns = int(os.environ.get("PDN_DECODE_SPLITS", "0")) or (1 if self.layers[0].attention.cache_k.shape[1] <= 256 else 4)
            st.update(packs=packs, graphs={}, nograph=False, host_pos=None, ns=ns,
                      ids=hp.zeros((B, 1), np.int64), pos=hp.zeros((1,), np.int32),
                      cand_v=hp.empty((B, nblk), np.float32), cand_i=hp.empty((B, nblk), np.int32),
                      # tokens by position: (*hist_ptr)[pos] is what the step at `pos` picked -- the array handed to
                      # the caller; a fresh history per generation (the pointer lives on the device, the graph holds
                      # only ITS address), so arrays returned earlier are never rewritten
                      hist_ptr=hp.zeros((1,), np.int64), hist=None,
                      **{n: hp.empty((B, w), np.float32) for n, w in
                         (("x", D), ("qkv", 3 * D), ("att", ns * H * (4 + D // H)), ("gu", 2 * F), ("logits", V))})
            # three launches per layer (csrc/decode_layer.hip): the output / down projections leave per-head /
            # per-32-hidden-unit records that the next kernel's staging adds to the residual row
            J = _lib.lib().query("pdn_decode_mlp_slices", F)
            st["fused"] = bool(Llama.fused_decode and J and D <= 1024 and ns * H <= 256 and len(self.layers) > 0 and all(
                l.ffn.gate.weight.data.is_contiguous() and l.ffn.up.weight.data.is_contiguous() for l in self.layers))
            # two launches per layer (csrc/decode_block.hip): the q | k | v projection inside the attention kernel, one
            # more record per head for the new key
            st["block"] = bool(st["fused"] and int(Llama.fused_decode) >= 2 and
                               _lib.lib().query("pdn_decode_block_supported", D, H, D // H, ns))
            # (block path: the number of key ranges follows the position -- 256 cached keys per range, one captured
            #  step per count -- unless PDN_DECODE_SPLITS pins it)
            cache_len = self.layers[0].attention.cache_k.shape[1]
            st["ns_max"] = ns if os.environ.get("PDN_DECODE_SPLITS") else min(7, max(1, -(-(cache_len - 1) // 256)))
            if st["block"] and not _lib.lib().query("pdn_decode_block_supported", D, H, D // H, st["ns_max"]):
                st["ns_max"] = ns
            # a workgroup of the block kernel holds the scores of ceil(cache_len / ranges) positions in LDS whatever the
            # position: `ns_min` = the fewest ranges a cache of this length allows (long caches start above one range);
            # none up to ns_max -> the three-launch path
            st["ns_min"] = 1
            if st["block"]:
                fits = [n for n in range(1, st["ns_max"] + 1)
                        if 0 < _lib.lib().query("pdn_decode_block_lds_bytes", D, H, D // H, n, cache_len) <= 64 * 1024]
                if fits:
                    st["ns_min"] = fits[0]
                else:
                    st["block"] = False
            if st["fused"]:
                st.update(J=J, recs=hp.empty((B, (max(ns, st["ns_max"]) + 1) * H * (4 + D)), np.float32), dparts=hp.empty((B, J * D), np.float32),
                          xa=hp.empty((B, D), np.float32), xb=hp.empty((B, D), np.float32))
            self._decode_ws = {"logits": st["logits"], "x": st["x"]}
        self._decode_st = st
        return st if ok else None

    def _decode_ns(self, st, pos):
        """Key ranges per head for the step at position `pos`."""
        if not st.get("block"):
            return st["ns"]
        if os.environ.get("PDN_DECODE_SPLITS"):
            return max(st["ns"], st.get("ns_min", 1))
        return min(st["ns_max"], max(st.get("ns_min", 1), -(-pos // 256)))

    def _decode_launches(self, st, ns=None):
        """The launches of one decode step (2 per layer + 2; 3 or 5 per layer at lower `fused_decode` levels); every argument is fixed for the lifetime of `st` (the position
        and the token ids are read from device memory), so the sequence can be captured once and replayed."""
        from .. import hipnp as hp, _lib
        L, s = _lib.lib(), hp.stream()
        D, H, F, V, B = self.embed_dim, self.n_heads, self.ffn_dim, self.vocab_size, st["B"]
        hd = D // H
        x, qkv, att, gu, logits = (st[n]._ptr for n in ("x", "qkv", "att", "gu", "logits"))
        pos = st["pos"]._ptr
        # (x = embedding rows of the current ids: left there by the previous step's pick kernel, or by
        #  `_decode_gather` when the ids came from outside)
        emb = self.tok_embedding.weight.data
        cos, sin = self.freqs_cos.data._ptr, self.freqs_sin.data._ptr
        head = self.lm_head
        bias = head.bias.data._ptr if getattr(head, "bias", None) is not None else None
        if st["fused"]:
            J, ns = st["J"], (st["ns"] if ns is None else ns)
            recs, dparts, xa, xb = (st[n]._ptr for n in ("recs", "dparts", "xa", "xb"))
            rrs = ns * H * (4 + D)
            for li, (layer, (wqkv, _)) in enumerate(zip(self.layers, st["packs"])):
                a, f = layer.attention, layer.ffn
                ck, cv = a.cache_k.data, a.cache_v.data
                nrm = layer.input_norm
                if st["block"]:
                    # x = previous block's h + its feed-forward records (-> xa); q | k | v, RoPE, cache append, attention
                    # and each head's rows of Wo in one launch: records of ns key ranges + the new key
                    L.call("pdn_decode_block_f32", x if li == 0 else xb, D, None if li == 0 else dparts, 0 if li == 0 else J,
                           J * D, xa, D, nrm.weight.data._ptr, nrm.eps, wqkv._ptr, D, wqkv._strides[0], cos, sin, ck._ptr,
                           cv._ptr, ck._strides[0], pos, ck.shape[1], a.O.weight.data._ptr, D, recs, B, H, hd, ns, s)
                    nrm = layer.post_attn_norm
                    L.call("pdn_decode_mlp_f32", xa, D, recs, (ns + 1) * H * (4 + D), ns + 1, H, xb, D,
                           nrm.weight.data._ptr, nrm.eps, f.gate.weight.data._ptr, f.up.weight.data._ptr, F,
                           f.down.weight.data._ptr, D, dparts, J * D, B, D, F, s)
                    continue
                # [q | k | v] = RMSNorm(x) @ [Wq | Wk | Wv]; x = previous block's h + its feed-forward records
                if li == 0:
                    L.call("pdn_decode_gemv_f32", x, D, nrm.weight.data._ptr, nrm.eps, wqkv._ptr, D, D, wqkv._strides[0],
                           None, None, 0, qkv, 3 * D, B, D, 3 * D, 0, 0, 0, None, None, s)
                else:
                    L.call("pdn_decode_gemv_sum_f32", xb, D, dparts, J, J * D, xa, D, nrm.weight.data._ptr, nrm.eps,
                           wqkv._ptr, D, D, wqkv._strides[0], None, qkv, 3 * D, B, D, 3 * D, None, None, s)
                # RoPE, cache append, attention over [0, pos], each head times its rows of Wo -> records
                L.call("pdn_decode_attention_oproj_f32", qkv, 3 * D, cos, sin, ck._ptr, cv._ptr, a.O.weight.data._ptr, D,
                       recs, B, H, hd, ns, ck._strides[0], pos, ck.shape[1], s)
                # h = x + merged records (-> xb); 32 hidden units per workgroup: gate | up, SwiGLU, their rows of Wdown
                nrm = layer.post_attn_norm
                L.call("pdn_decode_mlp_f32", x if li == 0 else xa, D, recs, rrs, ns, H, xb, D, nrm.weight.data._ptr,
                       nrm.eps, f.gate.weight.data._ptr, f.up.weight.data._ptr, F, f.down.weight.data._ptr, D, dparts,
                       J * D, B, D, F, s)
            L.call("pdn_decode_gemv_sum_f32", xb, D, dparts, J, J * D, None, 0, self.norm.weight.data._ptr, self.norm.eps,
                   head.weight.data._ptr, V, V, 0, bias, logits, V, B, D, V, st["cand_v"]._ptr, st["cand_i"]._ptr, s)
            L.call("pdn_decode_pick_tick_f32", st["cand_v"]._ptr, st["cand_i"]._ptr, B, st["cand_v"].shape[1],
                   st["ids"]._ptr, pos, st["hist_ptr"]._ptr, emb._ptr, emb._strides[0], D, x, s)
            return
        for layer, (wqkv, wgu) in zip(self.layers, st["packs"]):
            a, f = layer.attention, layer.ffn
            ck, cv = a.cache_k.data, a.cache_v.data
            cbs = ck._strides[0]
            # h = RMSNorm(x); [q | k | v] = h @ [Wq | Wk | Wv]
            L.call("pdn_decode_gemv_f32", x, D, layer.input_norm.weight.data._ptr, layer.input_norm.eps, wqkv._ptr, D, D,
                   wqkv._strides[0], None, None, 0, qkv, 3 * D, B, D, 3 * D, 0, 0, 0, None, None, s)
            # RoPE of q / k, cache append, attention over positions [0, pos]
            L.call("pdn_decode_attention_f32", qkv, 3 * D, cos, sin, ck._ptr, cv._ptr, att, B, H, hd, st["ns"], cbs, pos,
                   ck.shape[1], s)
            wo, wd = a.O.weight.data, f.down.weight.data
            # x += merge(att partials) @ Wo: the key-range partials are merged while the row is staged
            L.call("pdn_decode_gemv_f32", att, st["att"].shape[1], None, 0.0, wo._ptr, D, D, 0, None, x, D, x, D, B, D, D,
                   2, st["ns"], hd, None, None, s)
            L.call("pdn_decode_gemv_f32", x, D, layer.post_attn_norm.weight.data._ptr, layer.post_attn_norm.eps, wgu._ptr,
                   F, F, wgu._strides[0], None, None, 0, gu, 2 * F, B, D, 2 * F, 0, 0, 0, None, None, s)
            # x += (silu(gate) * up) @ Wdown: SwiGLU in the loads
            L.call("pdn_decode_gemv_f32", gu, 2 * F, None, 0.0, wd._ptr, D, D, 0, None, x, D, x, D, B, F, D, 1, 0, 0,
                   None, None, s)
        # vocabulary projection; every workgroup also leaves the first maximum of its columns, the pick kernel
        # finishes the argmax over those candidates (model.py:262-268) and advances the position
        L.call("pdn_decode_gemv_f32", x, D, self.norm.weight.data._ptr, self.norm.eps, head.weight.data._ptr, V, V, 0,
               bias, None, 0, logits, V, B, D, V, 0, 0, 0, st["cand_v"]._ptr, st["cand_i"]._ptr, s)
        L.call("pdn_decode_pick_tick_f32", st["cand_v"]._ptr, st["cand_i"]._ptr, B, st["cand_v"].shape[1],
               st["ids"]._ptr, pos, st["hist_ptr"]._ptr, emb._ptr, emb._strides[0], D, x, s)

    def _decode_gather(self, st):
        """x = tok_embedding[ids] for ids that did not come out of the previous step's pick kernel."""
        from .. import hipnp as hp, _lib
        emb = self.tok_embedding.weight.data
        _lib.lib().call("pdn_embedding_gather_f32", emb._ptr, self.vocab_size, self.embed_dim, emb._strides[0],
                        st["ids"]._ptr, st["B"], st["x"]._ptr, hp.err_flag_ptr(), hp.stream())

    def _decode_step_hip(self, ids, pos: int, more: bool = False):
        """One greedy decode step (one new token per sequence) without building tape nodes.  ids: (B, 1) int64
        device array; returns the next ids, (B, 1) int64.  The step is ONE hipGraph replay: norm + projection,
        RoPE + cache append, decode attention, SwiGLU + down projection and the greedy pick all read the position
        from device memory (csrc/decode.hip), so nothing changes between replays but the data."""
        from .. import hipnp as hp, _lib
        B = ids.shape[0]
        cache = self.layers[0].attention.cache_k
        # raw pointers / device-side offsets are formed from `pos`: refuse what the module path would also refuse
        # (the reference fails with a NumPy broadcast error, model.py:105-110)
        if pos < 0 or pos >= cache.shape[1] or pos >= self.freqs_cos.shape[0]:
            raise ValueError(f"decode position {pos} is outside the KV cache / RoPE table "
                             f"(max_seq_len {cache.shape[1]}, {self.freqs_cos.shape[0]} RoPE rows)")
        if B > cache.shape[0]:
            raise ValueError(f"batch {B} exceeds the KV cache's max_batch_size {cache.shape[0]}")
        st = self._decode_plan(B)
        if st is None:
            return self._decode_step_generic(ids, pos)
        ahead, st["ahead"] = st.get("ahead"), None
        if ahead is not None:
            if ahead[0] == pos and ahead[1] is ids:              # the step queued ahead is exactly this one
                out = st["last_out"] = ahead[2]
                if more and Llama.decode_ahead and pos + 1 < min(cache.shape[1], self.freqs_cos.shape[0]):
                    self._decode_ahead(st, pos + 1)
                return out
            hp.synchronize()                                     # a different request: the queued step is void
            st["host_pos"] = st["last_out"] = None               # (position and ids are uploaded again below)
        if st["host_pos"] != pos:
            st["pos"][...] = np.int32(pos)                       # (later steps: the device advances it itself)
            # a new generation: its own history -- slots in mapped host memory the pick kernel stores into directly
            st["hist"] = hp.Mailbox(cache.shape[1], (B, 1))
            st["hist_ptr"][...] = np.int64(st["hist"]._ptr)
        fresh = ids is not st["ids"] and ids is not st.get("last_out")
        if fresh:
            st["ids"][...] = ids                                 # (not the array the previous step returned: that
            self._decode_gather(st)                              # one's embedding row is already in x)
        ns = self._decode_ns(st, pos)
        g = False if st["nograph"] else st["graphs"].get(ns)
        if g is None and Llama.graph_decode and pos + 2 < min(cache.shape[1], self.freqs_cos.shape[0]):
            # capture once: hipnp.Graph runs the step twice for real (pool warm-up + first replay), which writes the
            # cache rows of positions pos and pos + 1 with exactly what the real steps will write there; the
            # position and the ids are then put back and the real step replayed
            # The pick kernel of those two runs stores its tokens into the history: it is pointed at a SCRATCH
            # history meanwhile, so that slots pos / pos + 1 of the real one stay "not written" (-1) until the real
            # steps store there (a later step with other ids would otherwise read the capture's token as its own).
            keep = st["ids"].copy()
            scratch = hp.Mailbox(cache.shape[1], (B, 1))
            st["hist_ptr"][...] = np.int64(scratch._ptr)
            try:
                g = hp.Graph()
                g.capture(lambda: self._decode_launches(st, ns))
                st["graphs"][ns] = g
            except _lib.HipLibraryError as e:
                if e.code != -2:                                 # PDN_EUNSUPPORTED: no graph support (the emulated
                    raise                                        # ABI) -> plain launches; anything else is a bug
                st["nograph"], g = True, False
            hp.synchronize()                                     # the capture's runs are done with the scratch history
            st["hist_ptr"][...] = np.int64(st["hist"]._ptr)
            st["pos"][...] = np.int32(pos)
            st["ids"][...] = keep
            self._decode_gather(st)
        if g:
            g.replay()
        else:
            self._decode_launches(st, ns)
        st["host_pos"] = pos + 1
        # the caller's own array = this position's slot of the history (host memory the GPU writes): reading the token
        # polls THAT slot only -- no copy command, no event -- while the compute stream may already run the next step
        out = st["last_out"] = st["hist"].slot(pos)
        if (more and Llama.decode_ahead and (st["graphs"] or st["nograph"])
                and pos + 1 < min(cache.shape[1], self.freqs_cos.shape[0])):
            self._decode_ahead(st, pos + 1)
        return out

    def _decode_ahead(self, st, pos):
        """Queue the step of position `pos` right behind the one just issued -- its input ids are already where the
        gather reads them -- so the GPU does not idle while the host hands the previous token to the caller.  The
        result is kept for the next `_decode_step_hip(last_out, pos)` call; any other call discards it."""
        ns = self._decode_ns(st, pos)
        g = False if st["nograph"] else st["graphs"].get(ns)
        if g is None:
            return                                               # (a new range count: its step is captured by the next call)
        if g:
            g.replay()
        else:
            self._decode_launches(st, ns)
        st["host_pos"] = pos + 1
        prev = st["last_out"]
        st["ahead"] = (pos, prev, st["hist"].slot(pos))

    def _decode_step_generic(self, ids, pos: int):
        """The same step from the library's generic entry points (skinny `pdn_gemm_f32`, RMSNorm, RoPE, decode
        attention, SwiGLU), ~77 launches from preallocated buffers: for shapes / layouts the graph path does not take."""
        from .. import hipnp as hp, _lib
        L, st = _lib.lib(), hp.stream()
        D, H, F, V = self.embed_dim, self.n_heads, self.ffn_dim, self.vocab_size
        hd, half = D // H, D // H // 2
        B = ids.shape[0]
        ws = getattr(self, "_decode_ws", None)
        if ws is None or ws["x"].device_index != hp._state["device"] or ws["x"].shape[0] != B:
            ws = {n: hp.empty((B, w), np.float32) for n, w in
                  (("x", D), ("h", D), ("q", D), ("att", D), ("g", F), ("u", F), ("sw", F), ("logits", V))}
            self._decode_ws = ws
        x, h, q, att, g, u, sw, logits = (ws[n]._ptr for n in ("x", "h", "q", "att", "g", "u", "sw", "logits"))

        def gemv(a_ptr, K, w, c_ptr, N, beta=0.0, bias=None, ldc=None):
            wd = w.data
            L.call("pdn_gemm_f32", B, N, K, 1.0, a_ptr, K, 1, wd._ptr, wd._strides[0], wd._strides[1], beta, c_ptr,
                   N if ldc is None else ldc, bias, 1, 1, 0, 0, 0, 0, 0, 0, None, None, 0, None, 0, st)

        emb = self.tok_embedding.weight.data
        idc = ids if ids.is_contiguous() else ids.copy()
        L.call("pdn_embedding_gather_f32", emb._ptr, V, D, emb._strides[0], idc._ptr, B, x, hp.err_flag_ptr(), st)
        cos = self.freqs_cos.data._ptr + pos * half * 4
        sin = self.freqs_sin.data._ptr + pos * half * 4
        for layer in self.layers:
            a, f = layer.attention, layer.ffn
            ck, cv = a.cache_k.data, a.cache_v.data
            cbs = ck._strides[0]                                          # floats between sequences in the cache
            kslot, vslot = ck._ptr + pos * D * 4, cv._ptr + pos * D * 4   # row b of the slot is cbs floats further
            L.call("pdn_rmsnorm_fwd_f32", x, layer.input_norm.weight.data._ptr, h, None, B, D, layer.input_norm.eps, st)
            gemv(h, D, a.Q.weight, q, D)
            gemv(h, D, a.K.weight, kslot, D, ldc=cbs)
            gemv(h, D, a.V.weight, vslot, D, ldc=cbs)
            L.call("pdn_rope_f32", q, cos, sin, q, B, 1, H, hd, 0, st)
            for b in range(B):
                L.call("pdn_rope_f32", kslot + b * cbs * 4, cos, sin, kslot + b * cbs * 4, 1, 1, H, hd, 0, st)
            L.call("pdn_attention_decode_f32", q, ck._ptr, cv._ptr, att, B, H, pos + 1, hd, cbs, st)
            gemv(att, D, a.O.weight, x, D, beta=1.0)                      # x += att @ Wo
            L.call("pdn_rmsnorm_fwd_f32", x, layer.post_attn_norm.weight.data._ptr, h, None, B, D,
                   layer.post_attn_norm.eps, st)
            gemv(h, D, f.gate.weight, g, F)
            gemv(h, D, f.up.weight, u, F)
            L.call("pdn_swiglu_fwd_f32", g, u, sw, B * F, st)
            gemv(sw, F, f.down.weight, x, D, beta=1.0)                    # x += swiglu @ Wdown
        L.call("pdn_rmsnorm_fwd_f32", x, self.norm.weight.data._ptr, h, None, B, D, self.norm.eps, st)
        gemv(h, D, self.lm_head.weight, logits, V,
             bias=self.lm_head.bias.data._ptr if getattr(self.lm_head, "bias", None) is not None else None)
        return ws["logits"].argmax(-1, keepdims=True)
