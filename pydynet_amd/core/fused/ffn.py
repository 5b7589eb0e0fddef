"""Feed-forward nodes: gate_up_swiglu and the one-node ffn_swiglu.
(One module of `pydynet_amd.core.fused`; the package docstring lists the reference chains each node replaces.)"""
from __future__ import annotations

import contextlib
import os

import numpy as np

from .norm import rms_norm
from ..tensor import _Operator
from ._common import (_hip, _L, _contig, _require_f32, _foldable, _beside, _is_leaf_f32, _pack_columns, _dx_of_shared_input)


class gate_up_swiglu(_Operator):
    """The FFN front end as ONE tape node (llm/llama/model.py:56-58): h = silu(x Wg) * (x Wu).
    Both bias-free projections write the halves of ONE packed (tokens, 2 * ffn) buffer (a single batched
    GEMM when the two weights are equally spaced in memory), the SwiGLU kernel reads the halves through
    a row stride.  Backward: d[gate | up] into one packed buffer, the two weight gradients as one
    batched GEMM and dx as ONE GEMM contracting over all 2 * ffn columns (the reference: 2 matmul + 5
    elementwise nodes forward, 2 separately accumulated input gradients backward)."""

    folds_existing = True
    enabled = True

    @staticmethod
    def applicable(x, wg, wu):
        return (gate_up_swiglu.enabled and x.device.is_hip and x.dtype == np.float32 and wg.dtype == np.float32
                and wu.dtype == np.float32 and wg.shape == wu.shape and wg.shape[1] % 4 == 0 and x.ndim >= 2)

    def __init__(self, x, w_gate, w_up):
        super().__init__(x, w_gate, w_up)

    @staticmethod
    def _halves(buf, T, F):
        hp = _hip()
        return hp.ndarray(buf._buf, buf._ptr, (2, T, F), (F, 2 * F, 1), buf.dtype)

    def forward_(self, x, wg, wu):
        _require_f32(self, x, wg, wu)
        hp, L = _hip(), _L()
        fin, F = wg.shape
        x2 = _contig(x.data).reshape(-1, fin)
        T = x2.shape[0]
        gu = hp.empty((T, 2 * F), np.float32)
        halves = self._halves(gu, T, F)
        ws = [_contig(wg.data), _contig(wu.data)]
        stack = hp.stacked_view(ws)
        if stack is not None:
            hp.gemm(x2, stack, halves)
        else:
            hp.gemm(x2, ws[0], halves[0])
            hp.gemm(x2, ws[1], halves[1])
        out = hp.empty(x.shape[:-1] + (F,), np.float32)
        L.call("pdn_swiglu_rows_fwd_f32", gu._ptr, out._ptr, T, F, hp.stream())
        self._saved = (x2, gu)
        return out

    def backward_all(self, dh):
        hp, L = _hip(), _L()
        x, wg, wu = self.last
        fin, F = wg.shape
        x2, gu = self._saved
        T = x2.shape[0]
        dh = _contig(dh)
        dgu = hp.empty((T, 2 * F), np.float32)
        L.call("pdn_swiglu_rows_bwd_f32", gu._ptr, dh._ptr, dgu._ptr, T, F, hp.stream())
        dhalves = self._halves(dgu, T, F)
        grads = [None] * 3
        weights = (wg, wu)
        gstack = None
        if all(w.requires_grad and _is_leaf_f32(w) for w in weights):
            gstack = hp.stacked_view([w.grad for w in weights])
        side = _beside(hp, fin, 2 * F, x.requires_grad and any(w.requires_grad for w in weights))
        dws = [hp.empty(w.shape, np.float32) if gstack is None and w.requires_grad and not _is_leaf_f32(w) else None
               for w in weights]
        with side or contextlib.nullcontext():
            if gstack is not None:
                hp.gemm(x2.T, dhalves, gstack, beta=1.0)
            else:
                for i, w in enumerate(weights):
                    if not w.requires_grad:
                        continue
                    if dws[i] is None:
                        hp.gemm(x2.T, dhalves[i], w.grad, beta=1.0)
                    else:
                        hp.gemm(x2.T, dhalves[i], dws[i])
                        grads[1 + i] = dws[i]
        if x.requires_grad:
            dx = hp.empty(x.shape, np.float32)
            ex = _foldable(self, 0, x)
            wcat = _pack_columns(hp, [wg.data, wu.data])                           # (fin, 2F)
            hp.gemm(dgu, wcat.T, dx.reshape(T, fin), residual=ex.reshape(T, fin) if ex is not None else None)
            grads[0] = dx
        if side is not None:
            side.join()
        return grads


class ffn_swiglu(_Operator):
    """The whole feed-forward block as ONE tape node (llm/llama/model.py:47-58):
    y = (silu(x Wg) * (x Wu)) Wd (+ residual).  What the merge buys over `gate_up_swiglu` + `linear` is that the two
    bandwidth passes of SwiGLU ride in GEMM epilogues (csrc/gemm_rowres.hip, round 4): forward, the packed gate | up
    projection writes h = silu(gate) * up beside [gate | up] in the same launch; backward, dh = dy Wd^T is never
    written -- the product's store reads the saved gate / up and leaves d[gate | up].  Shapes the epilogue kernels do
    not take (contraction other than 288, ffn not a multiple of 96, few rows) run the same algebra with the separate
    SwiGLU kernels.  The reference: 3 matmul + 5 elementwise nodes forward and their per-edge gradients."""

    folds_existing = True
    enabled = os.environ.get("PDN_NO_FFN_NODE", "0") != "1"            # (same-box A/B switches)
    epilogues = os.environ.get("PDN_NO_SWIGLU_EPILOGUE", "0") != "1"
    epilogue_min_rows = 4096         # below this the projections are launch-sized: the separate kernels are as good

    @staticmethod
    def applicable(x, wg, wu, wd):
        return (ffn_swiglu.enabled and gate_up_swiglu.applicable(x, wg, wu) and wd.dtype == np.float32
                and wd.shape == (wg.shape[1], wg.shape[0]))

    def __init__(self, x, w_gate, w_up, w_down, residual=None):
        self.has_res = residual is not None
        super().__init__(*([x, w_gate, w_up, w_down] + ([residual] if self.has_res else [])))

    @staticmethod
    def _epilogue(T, F, fin, stack):
        # (two arrays are always "equally spaced": the kernel's 32-bit offsets need them within 2^30 floats, not overlapping)
        return bool(ffn_swiglu.epilogues and stack is not None and T >= ffn_swiglu.epilogue_min_rows
                    and fin * F <= abs(stack._strides[0]) < (1 << 30) - fin * F
                    and _L().query("pdn_gateup_swiglu_supported", T, F, fin))

    def forward_(self, x, wg, wu, wd, r=None):
        _require_f32(self, x, wg, wu, wd, r)
        hp, L = _hip(), _L()
        fin, F = wg.shape
        T = x.size // fin
        gu = hp.empty((T, 2 * F), np.float32)
        h = hp.empty((T, F), np.float32)
        ws = [_contig(wg.data), _contig(wu.data)]
        stack = hp.stacked_view(ws)
        self.used_epilogue = self._epilogue(T, F, fin, stack)
        # other model widths: the same two epilogues in the stores of the tiled kernel (csrc/gemm.hip SWI)
        self.tiled_epilogue = bool(not self.used_epilogue and ffn_swiglu.epilogues and stack is not None
                                   and T >= ffn_swiglu.epilogue_min_rows and x.size == T * fin
                                   and L.query("pdn_gateup_swiglu_tiled_supported", T, F, fin))
        # a still-deferred RMSNorm in front (fused.rms_norm): its rows are normalised in the gate | up projection's A load
        self.norm_folded = bool(self.used_epilogue and isinstance(x, rms_norm) and x._pending is not None
                                and L.query("pdn_gateup_swiglu_norm_supported", T, F, fin))
        if self.norm_folded:
            raw_t, wn = x._pending
            raw = _contig(raw_t.data)
            xn, rms = hp.empty(x.shape, np.float32), hp.empty((T,), np.float32)
            L.call("pdn_gateup_swiglu_norm_fwd_f32", raw._ptr, _contig(wn.data)._ptr, x.eps, xn._ptr, rms._ptr, ws[0]._ptr,
                   (ws[1]._ptr - ws[0]._ptr) // 4, gu._ptr, h._ptr, T, F, fin, fin, hp.stream())
            x._adopt(raw, rms, xn)
        x2 = _contig(x.data).reshape(-1, fin)
        if self.norm_folded:
            pass
        elif self.used_epilogue:
            L.call("pdn_gateup_swiglu_fwd_f32", x2._ptr, ws[0]._ptr, (ws[1]._ptr - ws[0]._ptr) // 4, gu._ptr, h._ptr,
                   T, F, fin, fin, hp.stream())
        elif self.tiled_epilogue:
            wsp, wsb = hp.workspace(L.query("pdn_gateup_swiglu_tiled_workspace_bytes", F, fin))
            L.call("pdn_gateup_swiglu_tiled_fwd_f32", x2._ptr, x2._strides[0], ws[0]._ptr, (ws[1]._ptr - ws[0]._ptr) // 4,
                   gu._ptr, h._ptr, T, F, fin, wsp, wsb, hp.stream())
        else:
            halves = gate_up_swiglu._halves(gu, T, F)
            if stack is not None:
                hp.gemm(x2, stack, halves)
            else:
                hp.gemm(x2, ws[0], halves[0])
                hp.gemm(x2, ws[1], halves[1])
            L.call("pdn_swiglu_rows_fwd_f32", gu._ptr, h._ptr, T, F, hp.stream())
        out = hp.empty(x.shape[:-1] + (fin,), np.float32)
        res = _contig(r.data).reshape(-1, fin) if r is not None else None
        hp.gemm(h, wd.data, out.reshape(-1, fin), residual=res)
        self._saved = (x2, gu, h)
        return out

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, wg, wu, wd = self.last[:4]
        fin, F = wg.shape
        x2, gu, h = self._saved
        T = x2.shape[0]
        grads = [None] * len(self.last)
        if self.has_res and self.last[4].requires_grad:
            grads[4] = g
        g2 = _contig(g).reshape(T, fin)
        # down projection: dWd += h^T g
        if wd.requires_grad:
            if _is_leaf_f32(wd):
                hp.gemm(h.T, g2, wd.grad, beta=1.0)
            else:
                grads[3] = hp.empty((F, fin), np.float32)
                hp.gemm(h.T, g2, grads[3])
        if not (x.requires_grad or wg.requires_grad or wu.requires_grad):
            return grads
        # d[gate | up] = SwiGLU'(gate, up) o (g Wd^T)
        dgu = hp.empty((T, 2 * F), np.float32)
        wdd = _contig(wd.data)
        if self.used_epilogue:
            L.call("pdn_swiglu_bwd_gemm_f32", g2._ptr, wdd._ptr, gu._ptr, dgu._ptr, T, F, fin, fin, hp.stream())
        elif self.tiled_epilogue and L.query("pdn_swiglu_bwd_tiled_supported", T, F, fin):
            L.call("pdn_swiglu_bwd_tiled_f32", g2._ptr, g2._strides[0], wdd._ptr, gu._ptr, dgu._ptr, T, F, fin, hp.stream())
        else:
            dh = hp.empty((T, F), np.float32)
            hp.gemm(g2, wdd.T, dh)
            L.call("pdn_swiglu_rows_bwd_f32", gu._ptr, dh._ptr, dgu._ptr, T, F, hp.stream())
        dhalves = gate_up_swiglu._halves(dgu, T, F)
        weights = (wg, wu)
        gstack = None
        if all(w.requires_grad and _is_leaf_f32(w) for w in weights):
            gstack = hp.stacked_view([w.grad for w in weights])
        if gstack is not None:
            hp.gemm(x2.T, dhalves, gstack, beta=1.0)
        else:
            for i, w in enumerate(weights):
                if not w.requires_grad:
                    continue
                if _is_leaf_f32(w):
                    hp.gemm(x2.T, dhalves[i], w.grad, beta=1.0)
                else:
                    grads[1 + i] = hp.empty(w.shape, np.float32)
                    hp.gemm(x2.T, dhalves[i], grads[1 + i])
        if x.requires_grad:
            # (a residual that IS x hands its gradient g over separately: the engine adds it)
            grads[0] = _dx_of_shared_input(hp, self, x, dgu, (wg, wu), T, fin)
        return grads
