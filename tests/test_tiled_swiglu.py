"""SwiGLU in the stores of the tiled kernel (csrc/gemm.hip SWI instantiations): the two epilogues of `fused.ffn_swiglu`
for model widths other than 288 -- llm/llama/model.py:56-58 `silu(x Wg) * (x Wu)` forward, `dh = dy W_down^T` backward
(tensor.py:670) with the SwiGLU derivative applied in the store.

Kernel level against float64 NumPy statements of those lines (incl. an ffn width that is not a multiple of the 128 /
256-column tiles); node level: `fused.ffn_swiglu` at widths 128 / 512 against the same node with the epilogues off.
Tolerance 1e-4 relative to the tensor's largest entry.  Runs on the emulated C ABI and (-m gpu) on MI355X."""
import ctypes

import numpy as np
import pytest

import pydynet_amd as pdn
from pydynet_amd.core import fused
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


def _counters(L):
    buf = (ctypes.c_int64 * 21)()
    L.call("pdn_kernel_counters", buf, 21, 1)
    return list(buf)


def check_tiled_swiglu_kernels(device):
    L, hp = _lib_hp()
    for (M, F, K, seed) in ((4096, 1376, 512, 0), (4224, 256, 64, 1), (4096, 1024, 384, 2), (8192, 2048, 100, 3)):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((2, K, F)) / np.sqrt(K)).astype(np.float32)           # Wg, Wu equally spaced
        wd = (rng.standard_normal((F, K)) / np.sqrt(F)).astype(np.float32)
        dy = rng.standard_normal((M, K)).astype(np.float32)
        assert L.query("pdn_gateup_swiglu_tiled_supported", M, F, K) and L.query("pdn_swiglu_bwd_tiled_supported", M, F, K)
        xd, wdev, wdd, dyd = hp.from_numpy(x), hp.from_numpy(w), hp.from_numpy(wd), hp.from_numpy(dy)
        gu, h = hp.empty((M, 2 * F), np.float32), hp.empty((M, F), np.float32)
        wsp, wsb = hp.workspace(L.query("pdn_gateup_swiglu_tiled_workspace_bytes", F, K))
        _counters(L)
        L.call("pdn_gateup_swiglu_tiled_fwd_f32", xd._ptr, K, wdev._ptr, K * F, gu._ptr, h._ptr, M, F, K, wsp, wsb, hp.stream())
        g64 = x.astype(np.float64) @ w[0].astype(np.float64)
        u64 = x.astype(np.float64) @ w[1].astype(np.float64)
        close(gu, np.concatenate([g64, u64], 1), f"[gate | up] {M} x {F} x {K}")
        close(h, g64 / (1 + np.exp(-g64)) * u64, "silu(gate) * up")
        dgu = hp.empty((M, 2 * F), np.float32)
        L.call("pdn_swiglu_bwd_tiled_f32", dyd._ptr, K, wdd._ptr, gu._ptr, dgu._ptr, M, F, K, hp.stream())
        c = _counters(L)
        assert c[19] == 1 and c[20] == 1, c[19:21]
        dh = dy.astype(np.float64) @ wd.astype(np.float64).T
        gg, uu = host(gu)[:, :F].astype(np.float64), host(gu)[:, F:].astype(np.float64)
        sg = 1 / (1 + np.exp(-gg))
        close(dgu, np.concatenate([dh * uu * sg * (1 + gg * (1 - sg)), dh * gg * sg], 1), "d[gate | up]")


device_variants(globals(), check_tiled_swiglu_kernels)


def _ffn_step(device, dim, F, T, epilogues):
    import pydynet_amd.nn as nn
    saved = fused.ffn_swiglu.epilogues
    fused.ffn_swiglu.epilogues = epilogues
    try:
        rng = np.random.default_rng(5)
        mk = lambda *s: pdn.Tensor((rng.standard_normal(s) / np.sqrt(s[0])).astype(np.float32), device=device, requires_grad=True)
        # gate and up weights equally spaced in one buffer, as llm/llama lays them out
        wgu = pdn.Tensor((rng.standard_normal((2, dim, F)) / np.sqrt(dim)).astype(np.float32), device=device)
        wg, wu = pdn.Tensor(wgu.data[0], device=device, requires_grad=True), pdn.Tensor(wgu.data[1], device=device, requires_grad=True)
        wd = mk(F, dim)
        x = pdn.Tensor(rng.standard_normal((T, dim)).astype(np.float32), device=device, requires_grad=True)
        node = fused.ffn_swiglu(x, wg, wu, wd, residual=x)
        loss = (node * node).sum()
        loss.backward()
        return (float(loss.item()), [host(t.grad) for t in (x, wg, wu, wd)], getattr(node, "tiled_epilogue", False),
                getattr(node, "used_epilogue", False))
    finally:
        fused.ffn_swiglu.epilogues = saved


def check_ffn_node_other_widths(device):
    L, _ = _lib_hp()
    for dim, F in ((128, 256), (512, 1376)):
        _counters(L)
        l1, g1, tiled, rowres = _ffn_step(device, dim, F, 4096, True)
        c = _counters(L)
        assert tiled and not rowres and c[19] == 1 and c[20] == 1, (tiled, rowres, c[19:21])
        l0, g0, tiled0, _ = _ffn_step(device, dim, F, 4096, False)
        assert not tiled0
        assert abs(l1 - l0) <= RT * abs(l0)
        for a, b, nm in zip(g1, g0, ("x", "w_gate", "w_up", "w_down")):
            close(a, b, f"gradient of {nm}, dim {dim}")


device_variants(globals(), check_ffn_node_other_widths)
