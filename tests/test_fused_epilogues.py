"""Projections with the bandwidth pass next to them folded into the accumulator store (csrc/gemm_rowres.hip, round 4):

* `pdn_gateup_swiglu_fwd_f32`  -- x [Wg | Wu] with h = silu(gate) * up written beside [gate | up]
  (llm/llama/model.py:56-58, nn/functional.py:39-40),
* `pdn_swiglu_bwd_gemm_f32`    -- d[gate | up] straight from dy W_down^T, dh never stored,
* `pdn_qkv_rope_fwd_f32`       -- x [Wq | Wk | Wv] with RoPE on the q, k blocks (model.py:23-44, 93-104),
* `pdn_attention_bwd_rotated_f32` -- the attention backward for q, k that arrive rotated,

each against a float64 NumPy statement of the reference lines, and the tape nodes built on them (`fused.ffn_swiglu`,
`fused.qkv_attention` with the rotated projection) against the same modules with the epilogues switched off.
Tolerance 1e-4 relative to the tensor's largest entry (north_star).  Runs on the emulated C ABI and (-m gpu) on MI355X.
"""
import ctypes

import numpy as np
import pytest

import pydynet_amd as pdn
from pydynet_amd.core import fused
from pydynet_amd.core.tensor import Graph
from tests.conftest import device_variants

RT = 1e-4


def host(x):
    return x.numpy() if isinstance(x, pdn.Tensor) else (x if isinstance(x, np.ndarray) else x.get())


def close(a, b, what, rt=RT):
    a, b = np.asarray(host(a), np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a - b).max())
    assert err <= 1e-7 + rt * scale, (what, err, scale)


def _lib_hp():
    from pydynet_amd import _lib, hipnp
    return _lib.lib(), hipnp


def _silu(g):
    return g / (1 + np.exp(-g))


def _stack(hp, mats):
    """`mats` (equal shapes) in one buffer, equally spaced: (views, stride in floats)."""
    buf = hp.empty((len(mats),) + mats[0].shape, np.float32)
    views = []
    for i, m in enumerate(mats):
        buf[i] = hp.from_numpy(m)
        views.append(buf[i])
    return buf, views, int(np.prod(mats[0].shape))


# ---- kernel level ------------------------------------------------------------------------------------------------
def _gateup_case(M, F, seed, up_first=False):
    L, hp = _lib_hp()
    K = 288
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float32)
    wg = (0.08 * rng.standard_normal((K, F))).astype(np.float32)
    wu = (0.08 * rng.standard_normal((K, F))).astype(np.float32)
    assert L.query("pdn_gateup_swiglu_supported", M, F, K)
    if up_first:                      # the up matrix BELOW the gate matrix in memory (how two parameters created
        buf, (du, dg), stride = _stack(hp, [wu, wg])         # up-first happen to be laid out): negative stride
        stride = -stride
    else:
        buf, (dg, du), stride = _stack(hp, [wg, wu])
    xd = hp.from_numpy(x)
    gu, h = hp.empty((M, 2 * F), np.float32), hp.empty((M, F), np.float32)
    L.call("pdn_gateup_swiglu_fwd_f32", xd._ptr, dg._ptr, stride, gu._ptr, h._ptr, M, F, K, K, hp.stream())
    g64, u64 = x.astype(np.float64) @ wg, x.astype(np.float64) @ wu
    close(gu.get()[:, :F], g64, "gate")
    close(gu.get()[:, F:], u64, "up")
    close(h, _silu(g64) * u64, "h = silu(gate) * up")
    # backward half: dgu from dy W_down^T and the saved gate | up
    wd = (0.08 * rng.standard_normal((F, K))).astype(np.float32)
    dy = rng.standard_normal((M, K)).astype(np.float32)
    dgu = hp.empty((M, 2 * F), np.float32)
    dyd, wdd = hp.from_numpy(dy), hp.from_numpy(wd)          # (named: a temporary's buffer is freed before the call)
    L.call("pdn_swiglu_bwd_gemm_f32", dyd._ptr, wdd._ptr, gu._ptr, dgu._ptr, M, F, K, K, hp.stream())
    gs, us = gu.get()[:, :F].astype(np.float64), gu.get()[:, F:].astype(np.float64)
    dh = dy.astype(np.float64) @ wd.astype(np.float64).T
    s = 1 / (1 + np.exp(-gs))
    close(dgu.get()[:, :F], dh * us * s * (1 + gs * (1 - s)), "d gate")
    close(dgu.get()[:, F:], dh * gs * s, "d up")


def check_gateup_swiglu_full_blocks(dev):
    _gateup_case(512, 192, 0)


def check_gateup_swiglu_ragged_rows_ffn768(dev):
    _gateup_case(300, 768, 1)             # 300 rows: the last wave is partly, the one after it wholly outside M


def check_gateup_swiglu_up_matrix_first_in_memory(dev):
    _gateup_case(256, 96, 7, up_first=True)


def _rope_ref(y, cos, sin, L, hd):
    """model.py:23-44 on (M, D) rows at positions m % L, interleaved pairs."""
    M, D = y.shape
    pos = np.arange(M) % L
    yh = y.reshape(M, D // hd, hd // 2, 2)
    c, s = cos[pos][:, None, :], sin[pos][:, None, :]
    out = np.empty_like(yh)
    out[..., 0] = yh[..., 0] * c - yh[..., 1] * s
    out[..., 1] = yh[..., 0] * s + yh[..., 1] * c
    return out.reshape(M, D)


def _tables(L, hd):
    inv = 1.0 / (10000 ** (np.arange(0, hd, 2)[: hd // 2] / hd))
    fr = np.outer(np.arange(L), inv)
    return np.cos(fr).astype(np.float32), np.sin(fr).astype(np.float32)


def _qkv_rope_case(B, Lq, hd, seed):
    L, hp = _lib_hp()
    K = D = 288
    M = B * Lq
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float32)
    ws = [(0.08 * rng.standard_normal((K, D))).astype(np.float32) for _ in range(3)]
    cos, sin = _tables(Lq, hd)
    assert L.query("pdn_qkv_rope_supported", M, D, K, Lq, hd)
    buf, views, stride = _stack(hp, ws)
    tab = hp.empty((Lq, hd, 2), np.float32)
    cd, sd, xd = hp.from_numpy(cos), hp.from_numpy(sin), hp.from_numpy(x)
    L.call("pdn_rope_table_f32", cd._ptr, sd._ptr, tab._ptr, Lq, hd, hp.stream())
    t = tab.get()
    assert np.array_equal(t[:, 0::2, 0], cos) and np.array_equal(t[:, 1::2, 0], cos)
    assert np.array_equal(t[:, 0::2, 1], -sin) and np.array_equal(t[:, 1::2, 1], sin)
    qkv = hp.empty((M, 3 * D), np.float32)
    L.call("pdn_qkv_rope_fwd_f32", xd._ptr, views[0]._ptr, stride, qkv._ptr, tab._ptr, M, D, K, Lq, hd, K, hp.stream())
    got = qkv.get()
    x64 = x.astype(np.float64)
    c64, s64 = cos.astype(np.float64), sin.astype(np.float64)
    close(got[:, :D], _rope_ref(x64 @ ws[0], c64, s64, Lq, hd), "rotated q")
    close(got[:, D:2 * D], _rope_ref(x64 @ ws[1], c64, s64, Lq, hd), "rotated k")
    close(got[:, 2 * D:], x64 @ ws[2], "v (not rotated)")


def check_qkv_rope_hd48(dev):
    _qkv_rope_case(4, 64, 48, 2)


def check_qkv_rope_hd96_ragged_tail(dev):
    _qkv_rope_case(3, 32, 96, 3)          # 96 rows: waves past the end of M


def check_attention_bwd_rotated_equals_plain(dev):
    """Rotating q, k first and calling the `rotated` backward = the backward that rotates inside."""
    L, hp = _lib_hp()
    B, H, Lq, hd = 50, 6, 64, 48          # 300 heads: several per workgroup in the persistent kernels
    D = H * hd
    rng = np.random.default_rng(4)
    qkv = rng.standard_normal((B * Lq, 3 * D)).astype(np.float32)
    do = rng.standard_normal((B, Lq, H, hd)).astype(np.float32)
    cos, sin = _tables(Lq, hd)
    cd, sd = hp.from_numpy(cos), hp.from_numpy(sin)
    rot = qkv.copy()
    rot[:, :D] = _rope_ref(qkv[:, :D].astype(np.float64), cos, sin, Lq, hd)
    rot[:, D:2 * D] = _rope_ref(qkv[:, D:2 * D].astype(np.float64), cos, sin, Lq, hd)
    res = []
    for data, fwd_rope, name in ((qkv, True, "pdn_attention_bwd_f32"), (rot, False, "pdn_attention_bwd_rotated_f32")):
        a = hp.from_numpy(data)
        o, lse = hp.empty((B, Lq, H, hd), np.float32), hp.empty((B, H, Lq), np.float32)
        q, k, v = a._ptr, a._ptr + 4 * D, a._ptr + 8 * D
        L.call("pdn_attention_fwd_f32", q, k, v, o._ptr, lse._ptr, B, H, Lq, hd, 3 * D, Lq * 3 * D, D, Lq * D, 1,
               cd._ptr if fwd_rope else None, sd._ptr if fwd_rope else None, hp.stream())
        d = hp.empty((B * Lq, 3 * D), np.float32)
        dod = hp.from_numpy(do)
        ws, wsb = hp.workspace(L.query("pdn_attention_bwd_workspace_bytes", B, H, Lq))
        L.call(name, q, k, v, o._ptr, dod._ptr, lse._ptr, d._ptr, d._ptr + 4 * D, d._ptr + 8 * D, B, H, Lq,
               hd, 3 * D, Lq * 3 * D, D, Lq * D, 1, cd._ptr, sd._ptr, ws, wsb, hp.stream())
        res.append((o.get(), d.get()))
    close(res[1][0], res[0][0], "o")
    close(res[1][1], res[0][1], "dq | dk | dv", 2e-5)


def _blocks_dx_case(M, kb, nb, seed, reverse=False):
    """dX = [d_1 | ... | d_nb] [W_1 | ... | W_nb]^T + residual with the weights read where they live."""
    L, hp = _lib_hp()
    rng = np.random.default_rng(seed)
    assert L.query("pdn_gemm_outres_blocks_supported", M, kb, nb)
    ws = [(0.05 * rng.standard_normal((288, kb))).astype(np.float32) for _ in range(nb)]
    d = rng.standard_normal((M, nb * kb)).astype(np.float32)
    res = rng.standard_normal((M, 288)).astype(np.float32)
    buf, views, stride = _stack(hp, ws[::-1] if reverse else ws)
    if reverse:                                       # block i BELOW block i - 1 in memory: negative stride
        views, stride = views[::-1], -stride
    dd, rd, out = hp.from_numpy(d), hp.from_numpy(res), hp.empty((M, 288), np.float32)
    L.call("pdn_gemm_outres_blocks_nt_f32", dd._ptr, views[0]._ptr, stride, kb, nb, out._ptr, rd._ptr, M, nb * kb, 288,
           hp.stream())
    rows = np.r_[0:64, M // 2:M // 2 + 64, M - 64:M]          # (float64 on a sample of rows: the product is 14+ GFLOP)
    ref = res[rows].astype(np.float64)
    for i, w in enumerate(ws):
        ref += d[rows, i * kb:(i + 1) * kb].astype(np.float64) @ w.astype(np.float64).T
    close(out.get()[rows], ref, "dX over weight blocks")


def check_dx_over_qkv_weight_blocks(dev):
    _blocks_dx_case(57344 + 96, 288, 3, 11)           # 8-wave workgroups, ragged last row block


def check_dx_over_gate_up_weight_blocks_reversed(dev):
    _blocks_dx_case(28672, 768, 2, 12, reverse=True)  # 4-wave workgroups (K >= 1536), weights in descending order


def _linear_lse_case(M, V, seed, sample=None):
    """logits = x W + b with the rows' log-sum-exp from the same launch (transposed accumulators), then the loss from one
    gather per row -- against float64 (llm/llama/model.py:179 + nn/functional.py:364-381)."""
    L, hp = _lib_hp()
    K = 288
    rng = np.random.default_rng(seed)
    assert L.query("pdn_linear_lse_supported", M, V, K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    x[5] *= 6.0                                          # a row with large logits: the running maximum has to move
    w = (0.2 * rng.standard_normal((K, V))).astype(np.float32)
    b = rng.standard_normal(V).astype(np.float32)
    t = rng.integers(0, V, M)
    t[:3] = (0, V - 1, V // 2)
    xd, wd, bd, td = hp.from_numpy(x), hp.from_numpy(w), hp.from_numpy(b), hp.from_numpy(t.astype(np.int64))
    logits, lse = hp.empty((M, V), np.float32), hp.empty((M,), np.float32)
    L.call("pdn_linear_lse_fwd_f32", xd._ptr, wd._ptr, bd._ptr, logits._ptr, lse._ptr, M, V, K, K, V, V, hp.stream())
    loss_row, out = hp.empty((M,), np.float32), hp.empty((1,), np.float32)
    L.call("pdn_cross_entropy_from_lse_f32", logits._ptr, V, lse._ptr, td._ptr, M, V, 1, loss_row._ptr, out._ptr,
           hp.err_flag_ptr(), hp.stream())
    rows = np.arange(M) if sample is None else np.r_[0:sample, M // 2:M // 2 + sample, M - sample:M]
    z = x[rows].astype(np.float64) @ w.astype(np.float64) + b
    m = z.max(-1, keepdims=True)
    ref_lse = (m + np.log(np.exp(z - m).sum(-1, keepdims=True)))[:, 0]
    got = logits.get()
    close(got[rows], z, "logits")
    assert np.abs(lse.get()[rows] - ref_lse).max() <= 1e-5 * np.abs(ref_lse).max() + 1e-5
    ref_rows = ref_lse - z[np.arange(len(rows)), t[rows]]
    assert np.abs(loss_row.get()[rows] - ref_rows).max() <= 2e-5 * np.abs(ref_rows).max() + 2e-5
    if sample is None:
        assert abs(float(out.get()[0]) - ref_rows.mean()) <= 1e-5 * abs(ref_rows.mean())
    hp.check_index_errors()


def check_linear_lse_tail_chunk_and_ragged_rows(dev):
    _linear_lse_case(49152 + 40, 352, 21)              # 3 chunks + a 64-column tail; the last wave partly outside M


def check_linear_lse_single_tile_tail(dev):
    _linear_lse_case(49152, 128, 22)                   # 1 chunk + a 32-column tail


# ---- node level: one Llama block with the epilogues on / off --------------------------------------------------------
def _block_step(dev, epilogues):
    from pydynet_amd.llm.llama import Llama
    saved = (fused.ffn_swiglu.enabled, fused.ffn_swiglu.epilogue_min_rows, fused.qkv_attention.rope_epilogue,
             fused.qkv_attention.rope_min_rows)
    fused.ffn_swiglu.enabled = epilogues
    fused.ffn_swiglu.epilogue_min_rows = 32
    fused.qkv_attention.rope_epilogue = epilogues
    fused.qkv_attention.rope_min_rows = 32
    try:
        Graph.clear()
        np.random.seed(5)
        V, D, H, F, Lq, B = 64, 288, 6, 192, 64, 2
        model = Llama(V, D, H, F, Lq, B, 2, np.float32)
        rng = np.random.default_rng(6)
        model.tok_embedding.weight.data[...] = (0.5 * rng.standard_normal((V, D))).astype(np.float32)
        model.to(dev)
        ids = rng.integers(0, V, (B, Lq))
        tgt = rng.integers(0, V, (B, Lq))
        made = {"ffn": 0, "ffn_epi": 0, "rot": 0}
        ffn_init, qkv_fwd = fused.ffn_swiglu.forward_, fused.qkv_attention.forward_

        def ffn_spy(self, *a):
            out = ffn_init(self, *a)
            made["ffn"] += 1
            made["ffn_epi"] += bool(self.used_epilogue)
            return out

        def qkv_spy(self, *a):
            out = qkv_fwd(self, *a)
            made["rot"] += bool(self.rotated)
            return out

        fused.ffn_swiglu.forward_, fused.qkv_attention.forward_ = ffn_spy, qkv_spy
        try:
            loss = model.loss(ids, tgt)
            loss.backward()
        finally:
            fused.ffn_swiglu.forward_, fused.qkv_attention.forward_ = ffn_init, qkv_fwd
        grads = {n: host(p.grad) for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
        return float(host(loss)), grads, made
    finally:
        (fused.ffn_swiglu.enabled, fused.ffn_swiglu.epilogue_min_rows, fused.qkv_attention.rope_epilogue,
         fused.qkv_attention.rope_min_rows) = saved


def check_llama_block_epilogues_vs_separate_kernels(dev):
    l1, g1, m1 = _block_step(dev, True)
    l0, g0, m0 = _block_step(dev, False)
    assert m1 == {"ffn": 2, "ffn_epi": 2, "rot": 2}, m1     # both layers took the epilogue kernels
    assert m0 == {"ffn": 0, "ffn_epi": 0, "rot": 0}, m0
    assert abs(l1 - l0) <= RT * abs(l0), (l1, l0)
    assert g1.keys() == g0.keys() and len(g1) >= 20
    for n in g0:
        close(g1[n], g0[n], f"grad {n}")


# ---- RMSNorm folded into the projections' A load (round 5: csrc/gemm_rowtile.hip NORM, fused.rms_norm deferred) -------
def _norm_fold_case(M, Lq, seed):
    """`pdn_{qkv_rope,gateup_swiglu}_norm_fwd_f32` on raw rows vs the standalone RMSNorm kernel followed by the plain entry
    points (nn/modules/norm.py:245-248 then llm/llama/model.py:93-104 / 56-58), and vs float64."""
    L, hp = _lib_hp()
    K = D = 288
    F, hd, eps = 768, 48, 1e-6
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((M, K)) * rng.uniform(0.2, 3.0, (M, 1))).astype(np.float32)
    wn = rng.uniform(0.5, 1.5, K).astype(np.float32)
    assert L.query("pdn_qkv_rope_norm_supported", M, D, K, Lq, hd) and L.query("pdn_gateup_swiglu_norm_supported", M, F, K)
    xd, wnd = hp.from_numpy(x), hp.from_numpy(wn)
    # the separate path
    xn0, rms0 = hp.empty((M, K), np.float32), hp.empty((M,), np.float32)
    L.call("pdn_rmsnorm_fwd_f32", xd._ptr, wnd._ptr, xn0._ptr, rms0._ptr, M, K, eps, hp.stream())
    x64 = x.astype(np.float64)
    r64 = np.sqrt((x64 * x64).mean(-1) + eps)
    xn64 = x64 / r64[:, None] * wn
    close(xn0, xn64, "standalone rmsnorm")
    # q | k | v + RoPE
    wsq = [(0.08 * rng.standard_normal((K, D))).astype(np.float32) for _ in range(3)]
    buf, views, stride = _stack(hp, wsq)
    cos, sin = _tables(Lq, hd)
    tab = hp.empty((Lq, hd, 2), np.float32)
    cd, sd = hp.from_numpy(cos), hp.from_numpy(sin)
    L.call("pdn_rope_table_f32", cd._ptr, sd._ptr, tab._ptr, Lq, hd, hp.stream())
    q0, q1 = hp.empty((M, 3 * D), np.float32), hp.empty((M, 3 * D), np.float32)
    xn1, rms1 = hp.empty((M, K), np.float32), hp.empty((M,), np.float32)
    L.call("pdn_qkv_rope_fwd_f32", xn0._ptr, views[0]._ptr, stride, q0._ptr, tab._ptr, M, D, K, Lq, hd, K, hp.stream())
    L.call("pdn_qkv_rope_norm_fwd_f32", xd._ptr, wnd._ptr, eps, xn1._ptr, rms1._ptr, views[0]._ptr, stride, q1._ptr,
           tab._ptr, M, D, K, Lq, hd, K, hp.stream())
    close(xn1, xn64, "normalised rows left by the q|k|v projection", 2e-6)
    close(rms1, r64, "rms left by the q|k|v projection", 2e-6)
    close(q1, q0.get(), "q|k|v: folded vs separate norm", 5e-6)
    c64, s64 = cos.astype(np.float64), sin.astype(np.float64)
    close(q1.get()[:, :D], _rope_ref(xn64 @ wsq[0], c64, s64, Lq, hd), "rotated q vs float64")
    close(q1.get()[:, 2 * D:], xn64 @ wsq[2], "v vs float64")
    # gate | up + SwiGLU
    wg = (0.08 * rng.standard_normal((K, F))).astype(np.float32)
    wu = (0.08 * rng.standard_normal((K, F))).astype(np.float32)
    bufg, (dg, du), gstride = _stack(hp, [wg, wu])
    gu0, h0, gu1, h1 = (hp.empty((M, n), np.float32) for n in (2 * F, F, 2 * F, F))
    xn2, rms2 = hp.empty((M, K), np.float32), hp.empty((M,), np.float32)
    L.call("pdn_gateup_swiglu_fwd_f32", xn0._ptr, dg._ptr, gstride, gu0._ptr, h0._ptr, M, F, K, K, hp.stream())
    L.call("pdn_gateup_swiglu_norm_fwd_f32", xd._ptr, wnd._ptr, eps, xn2._ptr, rms2._ptr, dg._ptr, gstride, gu1._ptr, h1._ptr,
           M, F, K, K, hp.stream())
    close(xn2, xn64, "normalised rows left by the gate|up projection", 2e-6)
    close(rms2, r64, "rms left by the gate|up projection", 2e-6)
    close(gu1, gu0.get(), "gate|up: folded vs separate norm", 5e-6)
    close(h1, h0.get(), "h: folded vs separate norm", 5e-6)
    g64, u64 = xn64 @ wg, xn64 @ wu
    close(h1, _silu(g64) * u64, "h vs float64")


def check_norm_fold_full_row_blocks(dev):
    _norm_fold_case(2560, 64, 11)


def check_norm_fold_ragged_rows(dev):
    _norm_fold_case(2592, 32, 12)          # 2592 = 10 x 256 + 32: the GUARD instantiations


def _norm_fold_step(dev, fold):
    from pydynet_amd.llm.llama import Llama
    saved = (fused.rms_norm.fold, fused.rms_norm.fold_min_rows, fused.ffn_swiglu.epilogue_min_rows,
             fused.qkv_attention.rope_min_rows)
    fused.rms_norm.fold, fused.rms_norm.fold_min_rows = fold, 32
    fused.ffn_swiglu.epilogue_min_rows = fused.qkv_attention.rope_min_rows = 32
    adopted = {"n": 0}
    orig = fused.rms_norm._adopt

    def spy(self, *a):
        adopted["n"] += 1
        return orig(self, *a)
    fused.rms_norm._adopt = spy
    try:
        Graph.clear()
        np.random.seed(8)
        V, D, H, F, Lq, B = 64, 288, 6, 768, 64, 40        # 2560 tokens: the tile-piece projections take them
        model = Llama(V, D, H, F, Lq, B, 2, np.float32)
        rng = np.random.default_rng(9)
        model.tok_embedding.weight.data[...] = (0.5 * rng.standard_normal((V, D))).astype(np.float32)
        for n, p_ in model.named_parameters():
            if n.endswith("norm.weight"):                   # (ones by default: make the weight gradient path non-trivial)
                p_.data[...] = rng.uniform(0.5, 1.5, p_.shape).astype(np.float32)
        model.to(dev)
        ids, tgt = rng.integers(0, V, (B, Lq)), rng.integers(0, V, (B, Lq))
        loss = model.loss(ids, tgt)
        loss.backward()
        grads = {n: host(p_.grad) for n, p_ in model.named_parameters() if p_.requires_grad and p_.grad is not None}
        return float(host(loss)), grads, adopted["n"]
    finally:
        fused.rms_norm._adopt = orig
        (fused.rms_norm.fold, fused.rms_norm.fold_min_rows, fused.ffn_swiglu.epilogue_min_rows,
         fused.qkv_attention.rope_min_rows) = saved


def check_llama_block_norm_fold_vs_separate_norm(dev):
    """Two Llama blocks, one step: the four block norms ride in the q|k|v / gate|up projections (the final norm in front
    of lm_head keeps its own kernel); loss and every gradient -- the norm weights' included -- equal the unfolded step."""
    l1, g1, n1 = _norm_fold_step(dev, True)
    l0, g0, n0 = _norm_fold_step(dev, False)
    assert n1 == 4 and n0 == 0, (n1, n0)
    assert abs(l1 - l0) <= 1e-5 * abs(l0), (l1, l0)
    assert g1.keys() == g0.keys() and len(g1) >= 20
    for n in g0:
        close(g1[n], g0[n], f"grad {n}", 2e-5)


for _fn in [check_gateup_swiglu_full_blocks, check_gateup_swiglu_ragged_rows_ffn768,
            check_gateup_swiglu_up_matrix_first_in_memory, check_qkv_rope_hd48,
            check_qkv_rope_hd96_ragged_tail, check_attention_bwd_rotated_equals_plain,
            check_dx_over_qkv_weight_blocks, check_dx_over_gate_up_weight_blocks_reversed,
            check_linear_lse_tail_chunk_and_ragged_rows, check_linear_lse_single_tile_tail,
            check_llama_block_epilogues_vs_separate_kernels, check_norm_fold_full_row_blocks,
            check_norm_fold_ragged_rows, check_llama_block_norm_fold_vs_separate_norm]:
    device_variants(globals(), _fn)
