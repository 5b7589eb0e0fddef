"""Sequences of 512 / 768 / 1024 positions as 256-row block pairs on the persistent attention kernels
(csrc/attention_blocks.hip over csrc/attention_p.hip; llm/llama/model.py:112-121 at the lengths finetune.py:44 allows).

Against a float64 statement of those lines and their gradients: causal and full, q | k | v as column blocks of ONE packed
projection (the strides `fused.qkv_attention` hands over), and the backward that takes q, k ALREADY ROTATED and rotates
dq, dk back (RoPE rides in the projection's store, model.py:23-44) -- there the table rows of a query block and of a key
block differ.  The launch counters confirm one forward launch per (query block, key block <= query block) pair."""
import ctypes
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _counters(L):
    buf = (ctypes.c_int64 * 21)()
    L.call("pdn_kernel_counters", buf, 21, 1)
    return list(buf)


def _rope_tables(L, hd, rng):
    ang = rng.uniform(0, 2 * np.pi, (L, hd // 2))
    return np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)


def _rot(x, c, s, sign):
    """(B, L, H, hd): pairs (2i, 2i + 1) rotated by the angle of (position, i); sign = -1 rotates back."""
    xe, xo = x[..., 0::2], x[..., 1::2]
    c, s = c[None, :, None, :], sign * s[None, :, None, :]
    out = np.empty_like(x)
    out[..., 0::2] = xe * c - xo * s
    out[..., 1::2] = xe * s + xo * c
    return out


@pytest.mark.parametrize("B,H,L,causal", [(2, 3, 512, 1), (1, 2, 768, 1), (1, 2, 1024, 1), (2, 2, 512, 0), (1, 1, 768, 0)])
def test_block_pairs_forward_backward_packed(hip, B, H, L, causal):
    from pydynet_amd import _lib
    Lb = _lib.lib()
    hd, D = 48, H * 48
    rng = np.random.default_rng(L + causal)
    qkv = rng.standard_normal((B, L, 3 * D)).astype(np.float32)
    qkv[0, L // 3, D:D + hd] *= 5.0                                # a spiky key
    do = rng.standard_normal((B, L, H, hd)).astype(np.float32)
    QKV, DO = hip.from_numpy(qkv), hip.from_numpy(do)
    dqkv = hip.empty((B, L, 3 * D))
    o, lse = hip.empty((B, L, H, hd)), hip.empty((B, H, L))
    qp, kp, vp = QKV._ptr, QKV._ptr + 4 * D, QKV._ptr + 8 * D
    _counters(Lb)
    Lb.call("pdn_attention_fwd_f32", qp, kp, vp, o._ptr, lse._ptr, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, causal, None, None,
            hip.stream())
    nb = L // 256
    pairs = nb * (nb + 1) // 2 if causal else nb * nb
    c = _counters(Lb)
    assert c[7] == pairs and c[9] == 0, ("persistent forward launches / resident launches", c[7], c[9])
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, L, H, hd) for i in range(3))
    q64, k64, v64, g64 = (a.astype(np.float64).transpose(0, 2, 1, 3) for a in (q, k, v, do))
    s = q64 @ k64.swapaxes(-1, -2) / math.sqrt(hd)
    if causal:
        s = s + np.triu(np.full((L, L), -np.inf), 1)
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    p = e / e.sum(-1, keepdims=True)
    assert rel_err(o.get(), (p @ v64).transpose(0, 2, 1, 3)) < 2e-5
    assert np.allclose(lse.get(), (m + np.log(e.sum(-1, keepdims=True)))[..., 0], rtol=1e-5, atol=1e-5)
    ws, wsb = hip.workspace(Lb.query("pdn_attention_bwd_workspace_bytes", B, H, L))
    dp_, dk_, dv_ = dqkv._ptr, dqkv._ptr + 4 * D, dqkv._ptr + 8 * D
    Lb.call("pdn_attention_bwd_f32", qp, kp, vp, o._ptr, DO._ptr, lse._ptr, dp_, dk_, dv_, B, H, L, hd, 3 * D, L * 3 * D, D,
            L * D, causal, None, None, ws, wsb, hip.stream())
    c = _counters(Lb)
    assert c[8] == pairs and c[10] == 0, ("persistent backward launches / resident launches", c[8], c[10])
    dp = g64 @ v64.swapaxes(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / math.sqrt(hd)
    got = dqkv.get()
    gq, gk, gv = (got[..., i * D:(i + 1) * D].reshape(B, L, H, hd) for i in range(3))
    ref_q, ref_k, ref_v = ((ds @ k64).transpose(0, 2, 1, 3), (ds.swapaxes(-1, -2) @ q64).transpose(0, 2, 1, 3),
                           (p.swapaxes(-1, -2) @ g64).transpose(0, 2, 1, 3))
    assert rel_err(gv, ref_v) < 5e-5 and rel_err(gq, ref_q) < 5e-5 and rel_err(gk, ref_k) < 5e-5
    # q, k taken as ALREADY ROTATED, dq / dk rotated back with the table rows of THEIR block
    cs, sn = _rope_tables(L, hd, rng)
    C, S = hip.from_numpy(cs), hip.from_numpy(sn)
    dq2 = hip.empty((B, L, 3 * D))
    Lb.call("pdn_attention_bwd_rotated_f32", qp, kp, vp, o._ptr, DO._ptr, lse._ptr, dq2._ptr, dq2._ptr + 4 * D, dq2._ptr + 8 * D,
            B, H, L, hd, 3 * D, L * 3 * D, D, L * D, causal, C._ptr, S._ptr, ws, wsb, hip.stream())
    got2 = dq2.get()
    rq, rk, rv = (got2[..., i * D:(i + 1) * D].reshape(B, L, H, hd) for i in range(3))
    assert rel_err(rq, _rot(ref_q, cs.astype(np.float64), sn.astype(np.float64), -1.0)) < 5e-5
    assert rel_err(rk, _rot(ref_k, cs.astype(np.float64), sn.astype(np.float64), -1.0)) < 5e-5
    assert rel_err(rv, ref_v) < 5e-5
