"""The fused decode layer's C-ABI entries (csrc/decode_layer.hip, decode_stage.h) against float64 NumPy statements of
llm/llama/model.py:105-121 (attention with KV cache, one new token) and model.py:47-58 (feed-forward), chained the
way `Llama._decode_launches` chains them:  q|k|v -> attention + per-head output projection records -> merge + norm +
gate|up + SwiGLU + per-slice down records -> sum + norm + next projection.  Tolerance 1e-5 of the row's scale (fp32
sums in a different order than NumPy's)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rms(x, w, eps):
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * w


def _rope(v, c, s):                          # v (..., H, hd) interleaved pairs
    a = v.reshape(v.shape[:-1] + (v.shape[-1] // 2, 2))
    out = np.empty_like(a)
    out[..., 0] = a[..., 0] * c - a[..., 1] * s
    out[..., 1] = a[..., 0] * s + a[..., 1] * c
    return out.reshape(v.shape)


@pytest.mark.parametrize("B,D,H,F,ns,pos", [(1, 288, 6, 768, 1, 37), (1, 288, 6, 768, 4, 300), (2, 288, 6, 768, 4, 2),
                                            (3, 512, 8, 1024, 2, 129), (1, 1024, 16, 2048, 1, 5), (5, 64, 4, 96, 3, 1)])
def test_fused_decode_layer_chain(hip, B, D, H, F, ns, pos):
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(B * 1000 + D + pos)
    hd, maxL, eps = D // H, 512, 1e-6
    f32 = np.float32
    x = rng.standard_normal((B, D)).astype(f32)
    wqkv = (rng.standard_normal((3, D, D)) / math.sqrt(D)).astype(f32)
    wo = (rng.standard_normal((D, D)) / math.sqrt(D)).astype(f32)
    wg = (rng.standard_normal((D, F)) / math.sqrt(D)).astype(f32)
    wu = (rng.standard_normal((D, F)) / math.sqrt(D)).astype(f32)
    wd = (rng.standard_normal((F, D)) / math.sqrt(F)).astype(f32)
    wn = (rng.standard_normal((D, 4 * D)) / math.sqrt(D)).astype(f32)        # the next kernel's matrix
    n1, n2, n3 = (1 + 0.1 * rng.standard_normal((3, D))).astype(f32)
    kc = rng.standard_normal((B, maxL, H, hd)).astype(f32)
    vc = rng.standard_normal((B, maxL, H, hd)).astype(f32)
    ang = rng.uniform(0, 6.28, (maxL, hd // 2))
    cos, sin = np.cos(ang).astype(f32), np.sin(ang).astype(f32)

    # ---- float64 statement ----
    X = x.astype(np.float64)
    qkv = np.einsum("bk,jkn->bjn", _rms(X, n1, eps), wqkv.astype(np.float64))
    q = _rope(qkv[:, 0].reshape(B, H, hd), cos[pos].astype(np.float64), sin[pos].astype(np.float64))
    k = _rope(qkv[:, 1].reshape(B, H, hd), cos[pos].astype(np.float64), sin[pos].astype(np.float64))
    v = qkv[:, 2].reshape(B, H, hd)
    K = kc[:, :pos + 1].astype(np.float64); K[:, pos] = k
    Vv = vc[:, :pos + 1].astype(np.float64); Vv[:, pos] = v
    s = np.einsum("bhd,bthd->bht", q, K) / math.sqrt(hd)
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    att = np.einsum("bht,bthd->bhd", p, Vv).reshape(B, D)
    h = X + att @ wo.astype(np.float64)
    n = _rms(h, n2, eps)
    g, u = n @ wg.astype(np.float64), n @ wu.astype(np.float64)
    out = h + (g / (1 + np.exp(-g)) * u) @ wd.astype(np.float64)
    nxt = _rms(out, n3, eps) @ wn.astype(np.float64)

    # ---- the four launches ----
    J = L.query("pdn_decode_mlp_slices", F)
    assert J == F // 32
    dev = {n_: hip.from_numpy(a) for n_, a in dict(x=x, wqkv=wqkv, wo=wo, wg=wg, wu=wu, wd=wd, wn=wn, n1=n1, n2=n2, n3=n3,
                                                    kc=kc, vc=vc, cos=cos, sin=sin).items()}
    QKV, REC = hip.empty((B, 3 * D)), hip.empty((B, ns * H * (4 + D)))
    XB, XC, PARTS, Y = hip.empty((B, D)), hip.empty((B, D)), hip.empty((B, J * D)), hip.empty((B, 4 * D))
    POS = hip.from_numpy(np.array([pos], np.int32))
    st = hip.stream()
    L.call("pdn_decode_gemv_f32", dev["x"]._ptr, D, dev["n1"]._ptr, eps, dev["wqkv"]._ptr, D, D, D * D, None, None, 0,
           QKV._ptr, 3 * D, B, D, 3 * D, 0, 0, 0, None, None, st)
    L.call("pdn_decode_attention_oproj_f32", QKV._ptr, 3 * D, dev["cos"]._ptr, dev["sin"]._ptr, dev["kc"]._ptr,
           dev["vc"]._ptr, dev["wo"]._ptr, D, REC._ptr, B, H, hd, ns, maxL * D, POS._ptr, maxL, st)
    L.call("pdn_decode_mlp_f32", dev["x"]._ptr, D, REC._ptr, ns * H * (4 + D), ns, H, XB._ptr, D, dev["n2"]._ptr, eps,
           dev["wg"]._ptr, dev["wu"]._ptr, F, dev["wd"]._ptr, D, PARTS._ptr, J * D, B, D, F, st)
    L.call("pdn_decode_gemv_sum_f32", XB._ptr, D, PARTS._ptr, J, J * D, XC._ptr, D, dev["n3"]._ptr, eps, dev["wn"]._ptr,
           4 * D, 4 * D, 0, None, Y._ptr, 4 * D, B, D, 4 * D, None, None, st)

    def close(got, ref, what):
        err = np.abs(got.astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert err < 1e-5, (what, err)
    close(dev["kc"].get()[:, pos], k, "appended k")
    close(dev["vc"].get()[:, pos], v, "appended v")
    close(XB.get(), h, "h = x + attention")
    close(XC.get(), out, "block output")
    close(Y.get(), nxt, "next projection")
    # the records of a key range without keys are zeros with l = 0 (ns = 4 at pos = 2: the last range is empty)
    rec = REC.get().reshape(B, ns, H, 4 + D)
    chunk = -(-(pos + 1) // ns)
    for sp in range(ns):
        if sp * chunk >= pos + 1:
            assert np.all(rec[:, sp, :, 1] == 0) and np.all(rec[:, sp, :, 4:] == 0)


def test_fused_decode_mlp_without_records_and_rejections(hip):
    """records = NULL: h is the base row itself; shapes outside the kernel's reach are refused, not mangled."""
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    B, D, F, eps = 2, 96, 64, 1e-5
    x = rng.standard_normal((B, D)).astype(np.float32)
    wg, wu = (rng.standard_normal((2, D, F)) / 10).astype(np.float32)
    wd = (rng.standard_normal((F, D)) / 8).astype(np.float32)
    nw = np.ones(D, np.float32)
    X, WG, WU, WD, NW = map(hip.from_numpy, (x, wg, wu, wd, nw))
    J = F // 32
    P = hip.empty((B, J * D))
    L.call("pdn_decode_mlp_f32", X._ptr, D, None, 0, 0, 0, None, 0, NW._ptr, eps, WG._ptr, WU._ptr, F, WD._ptr, D,
           P._ptr, J * D, B, D, F, hip.stream())
    n = _rms(x.astype(np.float64), nw, eps)
    g, u = n @ wg, n @ wu
    ref = (g / (1 + np.exp(-g)) * u) @ wd
    got = P.get().reshape(B, J, D).astype(np.float64).sum(1)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5
    assert L.query("pdn_decode_mlp_slices", 48) == 0
    with pytest.raises(_lib.HipLibraryError):
        L.call("pdn_decode_mlp_f32", X._ptr, D, None, 0, 0, 0, None, 0, NW._ptr, eps, WG._ptr, WU._ptr, F, WD._ptr, D,
               P._ptr, J * D, B, D, 48, hip.stream())
    with pytest.raises(_lib.HipLibraryError):
        L.call("pdn_decode_gemv_sum_f32", X._ptr, 2048, P._ptr, J, J * 2048, None, 0, NW._ptr, eps, WD._ptr, D, D, 0, None,
               P._ptr, D, 1, 2048, D, None, None, hip.stream())


@pytest.mark.parametrize("B,D,H,F,ns,pos,n_prev", [(1, 288, 6, 768, 1, 37, 24), (1, 288, 6, 768, 4, 300, 24),
                                                   (2, 288, 6, 768, 4, 0, 0), (2, 288, 6, 768, 2, 1, 5),
                                                   (3, 512, 8, 1024, 2, 129, 32), (1, 1024, 16, 2048, 7, 260, 3),
                                                   (1, 288, 6, 768, 1, 400, 0)])
def test_decode_block_chain(hip, B, D, H, F, ns, pos, n_prev):
    """pdn_decode_block_f32 (hand-off sum + norm + q|k|v + RoPE + cache append + attention + per-head Wo records, with
    the new key as its own softmax partial) -> pdn_decode_mlp_f32 -> pdn_decode_gemv_sum_f32 against float64."""
    from pydynet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(B * 977 + D + pos + ns)
    hd, maxL, eps = D // H, 512, 1e-6
    assert L.query("pdn_decode_block_supported", D, H, hd, ns) == 1
    f32 = np.float32
    base = rng.standard_normal((B, D)).astype(f32)
    prev = (0.3 * rng.standard_normal((B, max(n_prev, 1), D))).astype(f32)
    wqkv = (rng.standard_normal((3, D, D)) / math.sqrt(D)).astype(f32)
    wo = (rng.standard_normal((D, D)) / math.sqrt(D)).astype(f32)
    wg = (rng.standard_normal((D, F)) / math.sqrt(D)).astype(f32)
    wu = (rng.standard_normal((D, F)) / math.sqrt(D)).astype(f32)
    wd = (rng.standard_normal((F, D)) / math.sqrt(F)).astype(f32)
    wn = (rng.standard_normal((D, 2 * D)) / math.sqrt(D)).astype(f32)
    n1, n2, n3 = (1 + 0.1 * rng.standard_normal((3, D))).astype(f32)
    kc = rng.standard_normal((B, maxL, H, hd)).astype(f32)
    vc = rng.standard_normal((B, maxL, H, hd)).astype(f32)
    ang = rng.uniform(0, 6.28, (maxL, hd // 2))
    cos, sin = np.cos(ang).astype(f32), np.sin(ang).astype(f32)

    X = base.astype(np.float64) + (prev[:, :n_prev].astype(np.float64).sum(1) if n_prev else 0)
    qkv = np.einsum("bk,jkn->bjn", _rms(X, n1, eps), wqkv.astype(np.float64))
    q = _rope(qkv[:, 0].reshape(B, H, hd), cos[pos].astype(np.float64), sin[pos].astype(np.float64))
    k = _rope(qkv[:, 1].reshape(B, H, hd), cos[pos].astype(np.float64), sin[pos].astype(np.float64))
    v = qkv[:, 2].reshape(B, H, hd)
    K = kc[:, :pos + 1].astype(np.float64); K[:, pos] = k
    Vv = vc[:, :pos + 1].astype(np.float64); Vv[:, pos] = v
    s = np.einsum("bhd,bthd->bht", q, K) / math.sqrt(hd)
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    att = np.einsum("bht,bthd->bhd", p, Vv).reshape(B, D)
    h = X + att @ wo.astype(np.float64)
    n = _rms(h, n2, eps)
    g, u = n @ wg.astype(np.float64), n @ wu.astype(np.float64)
    out = h + (g / (1 + np.exp(-g)) * u) @ wd.astype(np.float64)
    nxt = _rms(out, n3, eps) @ wn.astype(np.float64)

    J = L.query("pdn_decode_mlp_slices", F)
    dev = {n_: hip.from_numpy(a) for n_, a in dict(base=base, prev=prev, wqkv=wqkv, wo=wo, wg=wg, wu=wu, wd=wd, wn=wn, n1=n1,
                                                    n2=n2, n3=n3, kc=kc, vc=vc, cos=cos, sin=sin).items()}
    R = (ns + 1) * H * (4 + D)
    REC, XA, XB, XC = hip.empty((B, R)), hip.empty((B, D)), hip.empty((B, D)), hip.empty((B, D))
    PARTS, Y = hip.empty((B, J * D)), hip.empty((B, 2 * D))
    POS = hip.from_numpy(np.array([pos], np.int32))
    st = hip.stream()
    L.call("pdn_decode_block_f32", dev["base"]._ptr, D, dev["prev"]._ptr if n_prev else None, n_prev, max(n_prev, 1) * D,
           XA._ptr, D, dev["n1"]._ptr, eps, dev["wqkv"]._ptr, D, D * D, dev["cos"]._ptr, dev["sin"]._ptr, dev["kc"]._ptr,
           dev["vc"]._ptr, maxL * D, POS._ptr, maxL, dev["wo"]._ptr, D, REC._ptr, B, H, hd, ns, st)
    L.call("pdn_decode_mlp_f32", XA._ptr, D, REC._ptr, R, ns + 1, H, XB._ptr, D, dev["n2"]._ptr, eps, dev["wg"]._ptr,
           dev["wu"]._ptr, F, dev["wd"]._ptr, D, PARTS._ptr, J * D, B, D, F, st)
    L.call("pdn_decode_gemv_sum_f32", XB._ptr, D, PARTS._ptr, J, J * D, XC._ptr, D, dev["n3"]._ptr, eps, dev["wn"]._ptr,
           2 * D, 2 * D, 0, None, Y._ptr, 2 * D, B, D, 2 * D, None, None, st)

    def close(got, ref, what):
        err = np.abs(got.astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert err < 1e-5, (what, err)
    close(XA.get(), X, "x = base + records")
    close(dev["kc"].get()[:, pos], k, "appended k")
    close(dev["vc"].get()[:, pos], v, "appended v")
    close(XB.get(), h, "h = x + attention")
    close(XC.get(), out, "block output")
    close(Y.get(), nxt, "next projection")
    rec = REC.get().reshape(B, ns + 1, H, 4 + D)
    assert np.all(rec[:, ns, :, 1] == 1.0)                          # the new key's own partial
    chunk = -(-pos // ns)
    for sp in range(ns):
        if sp * chunk >= pos:
            assert np.all(rec[:, sp, :, 1] == 0) and np.all(rec[:, sp, :, 4:] == 0)


def test_decode_block_rejects_other_head_dims(hip):
    from pydynet_amd import _lib
    L = _lib.lib()
    assert L.query("pdn_decode_block_supported", 256, 8, 32, 1) == 0
    assert L.query("pdn_decode_block_supported", 288, 6, 48, 8) == 0
    x = hip.empty((1, 256))
    with pytest.raises(_lib.HipLibraryError) as e:
        L.call("pdn_decode_block_f32", x._ptr, 256, None, 0, 256, x._ptr, 256, x._ptr, 1e-6, x._ptr, 256, 65536, x._ptr,
               x._ptr, x._ptr, x._ptr, 256, x._ptr, 1, x._ptr, 256, x._ptr, 1, 8, 32, 1, hip.stream())
    assert e.value.code == -2
