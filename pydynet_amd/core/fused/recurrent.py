"""Recurrent nodes: rnn_cell, lstm_cell, gru_cell, gru_sequence (persistent kernel).
(One module of `pydynet_amd.core.fused`; the package docstring lists the reference chains each node replaces.)"""
from __future__ import annotations

import numpy as np

from ..tensor import _Operator
from ._common import _hip, _L, _contig, _require_f32, _is_leaf_f32, _gemm_raw


def _cell_grads(node, hp, x, h, wx, wh, bias, xd, hd, dlin, grads, ix=0, ih=1, iwx=2, iwh=3, ib=4):
    """Gradients every recurrent cell shares once d(pre-activation) is known: dx = dlin Wx^T,
    dh = dlin Wh^T, dWx += x^T dlin, dWh += h^T dlin, db = column sums of dlin."""
    if x.requires_grad:
        dx = hp.empty(x.shape, np.float32)
        hp.gemm(dlin, wx.data.T, dx)
        grads[ix] = dx
    if h.requires_grad:
        dh = hp.empty(h.shape, np.float32)
        hp.gemm(dlin, wh.data.T, dh)
        grads[ih] = dh if grads[ih] is None else grads[ih] + dh
    for idx, a, w in ((iwx, xd, wx), (iwh, hd, wh)):
        if not w.requires_grad:
            continue
        if _is_leaf_f32(w):
            hp.gemm(a.T, dlin, w.grad, beta=1.0)
        else:
            dw = hp.empty(w.shape, np.float32)
            hp.gemm(a.T, dlin, dw)
            grads[idx] = dw
    if bias is not None and bias.requires_grad:
        grads[ib] = dlin.sum(0).reshape(bias.shape)


class rnn_cell(_Operator):
    """One Elman step (nn/modules/rnn.py:35-47) as a single tape node on the HIP device:
    h' = act(x Wx + h Wh + b): 2 GEMMs (the second accumulating, bias in the first's epilogue) + one
    pointwise kernel forward; one pointwise kernel + 4 GEMMs backward.  act: "tanh" | "relu"."""

    def __init__(self, x, h, wx, wh, bias=None, nonlinearity="tanh"):
        self.act = {"tanh": 0, "relu": 1}[nonlinearity]
        self.has_bias = bias is not None
        super().__init__(*((x, h, wx, wh) + ((bias,) if self.has_bias else ())))

    def forward_(self, x, h, wx, wh, bias=None):
        if self.xp is np:
            raise NotImplementedError("rnn_cell is the HIP fused path; the NumPy device composes generic ops")
        _require_f32(self, x, h, wx, wh, bias)
        hp, L = _hip(), _L()
        B, H = h.shape
        xd, hd = _contig(x.data), _contig(h.data)
        lin = hp.empty((B, H), np.float32)
        hp.gemm(xd, wx.data, lin, bias=bias.data.reshape(-1) if bias is not None else None)
        hp.gemm(hd, wh.data, lin, beta=1.0)
        y = hp.empty((B, H), np.float32)
        L.call("pdn_rnn_cell_fwd_f32", lin._ptr, y._ptr, y.size, self.act, hp.stream())
        self._saved = (xd, hd, lin)
        return y

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, h, wx, wh = self.last[:4]
        bias = self.last[4] if self.has_bias else None
        xd, hd, lin = self._saved
        g = _contig(g)
        dlin = hp.empty(lin.shape, np.float32)
        L.call("pdn_rnn_cell_bwd_f32", lin._ptr, self.data._ptr, g._ptr, dlin._ptr, dlin.size, self.act, hp.stream())
        grads = [None] * len(self.last)
        _cell_grads(self, hp, x, h, wx, wh, bias, xd, hd, dlin, grads)
        return grads


class lstm_cell(_Operator):
    """One LSTM step (nn/modules/rnn.py:244-262) as a single tape node on the HIP device.  The node's value
    is the packed pair (B, 2H) = [h' | c'] (the module hands out the two halves as views); 2 GEMMs + one
    pointwise kernel forward (12 generic nodes in the reference), one pointwise kernel + 4 GEMMs backward."""

    def __init__(self, x, h, c, wx, wh, bias=None):
        self.has_bias = bias is not None
        super().__init__(*((x, h, c, wx, wh) + ((bias,) if self.has_bias else ())))

    def forward_(self, x, h, c, wx, wh, bias=None):
        if self.xp is np:
            raise NotImplementedError("lstm_cell is the HIP fused path; the NumPy device composes generic ops")
        _require_f32(self, x, h, c, wx, wh, bias)
        hp, L = _hip(), _L()
        B, H = h.shape
        xd, hd, cd = _contig(x.data), _contig(h.data), _contig(c.data)
        lin = hp.empty((B, 4 * H), np.float32)
        hp.gemm(xd, wx.data, lin, bias=bias.data.reshape(-1) if bias is not None else None)
        hp.gemm(hd, wh.data, lin, beta=1.0)
        gates, tc, hc = hp.empty((B, 4 * H), np.float32), hp.empty((B, H), np.float32), hp.empty((B, 2 * H), np.float32)
        L.call("pdn_lstm_cell_fwd_f32", lin._ptr, cd._ptr, gates._ptr, tc._ptr, hc._ptr, B, H, hp.stream())
        self._saved = (xd, hd, cd, gates, tc)
        return hc

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, h, c, wx, wh = self.last[:5]
        bias = self.last[5] if self.has_bias else None
        xd, hd, cd, gates, tc = self._saved
        B, H = hd.shape
        g = _contig(g)
        dlin, dc = hp.empty((B, 4 * H), np.float32), hp.empty((B, H), np.float32)
        L.call("pdn_lstm_cell_bwd_f32", g._ptr, gates._ptr, tc._ptr, cd._ptr, dlin._ptr, dc._ptr, B, H, hp.stream())
        grads = [None] * len(self.last)
        if c.requires_grad:
            grads[2] = dc
        _cell_grads(self, hp, x, h, wx, wh, bias, xd, hd, dlin, grads, ix=0, ih=1, iwx=3, iwh=4, ib=5)
        return grads


class gru_cell(_Operator):
    """One GRU step (nn/modules/rnn.py:537-544) as a single tape node on the HIP device:
        [z, r] = sigmoid(x Wx1 + h Wh1 + b1);  n = tanh(x Wx2 + (r*h) Wh2 + b2);  h' = (1-z) h + z n
    4 GEMMs + 2 gate kernels forward, 8 GEMMs + 2 gate kernels backward (the generic composition is
    ~20 nodes / ~45 launches per step).  Inputs: x (B, in), h (B, H), Wx1, Wh1, Wx2, Wh2[, b1, b2]."""

    def __init__(self, x, h, wx1, wh1, wx2, wh2, b1=None, b2=None):
        self.has_bias = b1 is not None
        super().__init__(*((x, h, wx1, wh1, wx2, wh2) + ((b1, b2) if self.has_bias else ())))

    def forward_(self, x, h, wx1, wh1, wx2, wh2, b1=None, b2=None):
        if self.xp is np:
            raise NotImplementedError("gru_cell is the HIP fused path; the NumPy device composes generic ops")
        _require_f32(self, x, h, wx1, wh1, wx2, wh2, b1, b2)
        hp, L = _hip(), _L()
        B, H = h.shape
        xd, hd = _contig(x.data), _contig(h.data)
        g1 = hp.empty((B, 2 * H), np.float32)
        hp.gemm(xd, wx1.data, g1, bias=b1.data.reshape(-1) if b1 is not None else None)
        hp.gemm(hd, wh1.data, g1, beta=1.0)
        z, r, rh = (hp.empty((B, H), np.float32) for _ in range(3))
        L.call("pdn_gru_gates_fwd_f32", g1._ptr, hd._ptr, z._ptr, r._ptr, rh._ptr, B, H, hp.stream())
        g2 = hp.empty((B, H), np.float32)
        hp.gemm(xd, wx2.data, g2, bias=b2.data.reshape(-1) if b2 is not None else None)
        hp.gemm(rh, wh2.data, g2, beta=1.0)
        n, hn = hp.empty((B, H), np.float32), hp.empty((B, H), np.float32)
        L.call("pdn_gru_out_fwd_f32", g2._ptr, z._ptr, hd._ptr, n._ptr, hn._ptr, B, H, hp.stream())
        self._saved = (xd, hd, z, r, rh, n)
        return hn

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, h, wx1, wh1, wx2, wh2 = self.last[:6]
        b1, b2 = (self.last[6], self.last[7]) if self.has_bias else (None, None)
        xd, hd, z, r, rh, n = self._saved
        B, H = hd.shape
        g = _contig(g)
        dg2, dg1, dh = hp.empty((B, H), np.float32), hp.empty((B, 2 * H), np.float32), hp.empty((B, H), np.float32)
        L.call("pdn_gru_out_bwd_f32", g._ptr, z._ptr, n._ptr, hd._ptr, dg2._ptr, dg1._ptr, dh._ptr, B, H, hp.stream())
        drh = hp.empty((B, H), np.float32)
        hp.gemm(dg2, wh2.data.T, drh)
        L.call("pdn_gru_gates_bwd_f32", drh._ptr, r._ptr, hd._ptr, dg1._ptr, dh._ptr, B, H, hp.stream())
        grads = [None] * len(self.last)
        if h.requires_grad:
            hp.gemm(dg1, wh1.data.T, dh, beta=1.0)
            grads[1] = dh
        if x.requires_grad:
            dx = hp.empty(x.shape, np.float32)
            hp.gemm(dg2, wx2.data.T, dx)
            hp.gemm(dg1, wx1.data.T, dx, beta=1.0)
            grads[0] = dx
        for idx, (a, d, w) in enumerate(((xd, dg1, wx1), (hd, dg1, wh1), (xd, dg2, wx2), (rh, dg2, wh2)), start=2):
            if not w.requires_grad:
                continue
            if _is_leaf_f32(w):
                hp.gemm(a.T, d, w.grad, beta=1.0)
            else:
                dw = hp.empty(w.shape, np.float32)
                hp.gemm(a.T, d, dw)
                grads[idx] = dw
        if self.has_bias:
            if b1.requires_grad:
                grads[6] = dg1.sum(0).reshape(b1.shape)
            if b2.requires_grad:
                grads[7] = dg2.sum(0).reshape(b2.shape)
        return grads


class gru_sequence(_Operator):
    """A whole single-layer GRU over T steps as ONE tape node (nn/modules/rnn.py:537-544, 640-694):
    the input projections of all steps are hoisted into two GEMMs over (T*B, in); each step then costs
    two (B, H) x (H, .) GEMMs (the hoisted term rides in as the epilogue residual) and the two gate
    kernels; backward walks the steps in reverse with five launches each and forms every weight
    gradient with ONE long-K GEMM over the stacked per-step quantities.  The reference runs ~20 tape
    nodes per step.  Inputs: x (T, B, in), h0 (B, H), Wx1, Wh1, Wx2, Wh2[, b1, b2]; output (T, B, H)."""

    use_persistent = True      # class switch: False keeps the per-step launches (tests, A/B)

    def __init__(self, x, h0, wx1, wh1, wx2, wh2, b1=None, b2=None):
        self.has_bias = b1 is not None
        super().__init__(*((x, h0, wx1, wh1, wx2, wh2) + ((b1, b2) if self.has_bias else ())))

    def forward_(self, x, h0, wx1, wh1, wx2, wh2, b1=None, b2=None):
        if self.xp is np:
            raise NotImplementedError("gru_sequence is the HIP fused path")
        _require_f32(self, x, h0, wx1, wh1, wx2, wh2, b1, b2)
        hp, L = _hip(), _L()
        T, B, I = x.shape
        H = h0.shape[-1]
        x2 = _contig(x.data).reshape(T * B, I)
        g1x, g2x = hp.empty((T, B, 2 * H), np.float32), hp.empty((T, B, H), np.float32)
        hp.gemm(x2, wx1.data, g1x.reshape(T * B, 2 * H), bias=b1.data.reshape(-1) if b1 is not None else None)
        hp.gemm(x2, wx2.data, g2x.reshape(T * B, H), bias=b2.data.reshape(-1) if b2 is not None else None)
        out = hp.empty((T, B, H), np.float32)
        Z, R, RH, N = (hp.empty((T, B, H), np.float32) for _ in range(4))
        h0d = _contig(h0.data)
        st = hp.stream()
        self._persistent = bool(gru_sequence.use_persistent and L.query("pdn_gru_seq_supported", H))
        if self._persistent:
            # the whole time loop in ONE launch: a wave owns 32 sequences, h stays in its registers
            L.call("pdn_gru_seq_fwd_f32", g1x._ptr, g2x._ptr, h0d._ptr, _contig(wh1.data)._ptr, _contig(wh2.data)._ptr,
                   Z._ptr, R._ptr, RH._ptr, N._ptr, out._ptr, T, B, H, st)
            self._saved = (x2, h0d, Z, R, RH, N)
            return out
        g1, g2 = hp.empty((B, 2 * H), np.float32), hp.empty((B, H), np.float32)
        wh1d, wh2d = wh1.data, wh2.data
        sH, s2H = B * H * 4, B * 2 * H * 4                           # bytes per time step
        hprev = h0d._ptr
        for t in range(T):
            z, r, rh, n, o = Z._ptr + t * sH, R._ptr + t * sH, RH._ptr + t * sH, N._ptr + t * sH, out._ptr + t * sH
            _gemm_raw(L, st, B, 2 * H, H, hprev, H, 1, wh1d, g1._ptr, 2 * H, residual_ptr=g1x._ptr + t * s2H)
            L.call("pdn_gru_gates_fwd_f32", g1._ptr, hprev, z, r, rh, B, H, st)
            _gemm_raw(L, st, B, H, H, rh, H, 1, wh2d, g2._ptr, H, residual_ptr=g2x._ptr + t * sH)
            L.call("pdn_gru_out_fwd_f32", g2._ptr, z, hprev, n, o, B, H, st)
            hprev = o
        self._saved = (x2, h0d, Z, R, RH, N)
        return out

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, h0, wx1, wh1, wx2, wh2 = self.last[:6]
        b1, b2 = (self.last[6], self.last[7]) if self.has_bias else (None, None)
        x2, h0d, Z, R, RH, N = self._saved
        T, B, H = Z.shape
        I = x2.shape[1]
        g = _contig(g)
        out, st = self.data, hp.stream()
        dG1, dG2 = hp.empty((T, B, 2 * H), np.float32), hp.empty((T, B, H), np.float32)
        if self._persistent:
            dh = hp.empty((B, H), np.float32)
            L.call("pdn_gru_seq_bwd_f32", g._ptr, Z._ptr, R._ptr, N._ptr, out._ptr, h0d._ptr, _contig(wh1.data)._ptr,
                   _contig(wh2.data)._ptr, dG1._ptr, dG2._ptr, dh._ptr, T, B, H, st)
        else:
            dh = self._backward_steps(hp, L, st, g, out, h0d, Z, R, N, dG1, dG2, wh1.data, wh2.data, T, B, H)
        return self._finish_backward(hp, x, h0, wx1, wh1, wx2, wh2, b1, b2, x2, h0d, out, RH, dG1, dG2, dh, T, B, H, I)

    @staticmethod
    def _backward_steps(hp, L, st, g, out, h0d, Z, R, N, dG1, dG2, wh1d, wh2d, T, B, H):
        dh, dh2, drh = hp.zeros((B, H), np.float32), hp.empty((B, H), np.float32), hp.empty((B, H), np.float32)
        sH, s2H = B * H * 4, B * 2 * H * 4
        for t in range(T - 1, -1, -1):
            hprev = out._ptr + (t - 1) * sH if t > 0 else h0d._ptr
            z, r, n = Z._ptr + t * sH, R._ptr + t * sH, N._ptr + t * sH
            dg1, dg2 = dG1._ptr + t * s2H, dG2._ptr + t * sH
            # dh += g[t]: gradient of h_t = direct + from step t+1
            dh += g[t]
            L.call("pdn_gru_out_bwd_f32", dh._ptr, z, n, hprev, dg2, dg1, dh2._ptr, B, H, st)
            _gemm_raw(L, st, B, H, H, dg2, H, 1, wh2d, drh._ptr, H, b_transposed=True)
            L.call("pdn_gru_gates_bwd_f32", drh._ptr, r, hprev, dg1, dh2._ptr, B, H, st)
            _gemm_raw(L, st, B, H, 2 * H, dg1, 2 * H, 1, wh1d, dh2._ptr, H, beta=1.0, b_transposed=True)
            dh, dh2 = dh2, dh
        return dh

    def _finish_backward(self, hp, x, h0, wx1, wh1, wx2, wh2, b1, b2, x2, h0d, out, RH, dG1, dG2, dh, T, B, H, I):
        grads = [None] * len(self.last)
        if h0.requires_grad:
            grads[1] = dh
        d1, d2 = dG1.reshape(T * B, 2 * H), dG2.reshape(T * B, H)
        if x.requires_grad:
            dx = hp.empty(x.shape, np.float32)
            hp.gemm(d1, wx1.data.T, dx.reshape(T * B, I))
            hp.gemm(d2, wx2.data.T, dx.reshape(T * B, I), beta=1.0)
            grads[0] = dx
        hprev_all = hp.empty((T, B, H), np.float32)                      # h_{t-1} for every step, stacked
        hprev_all[0] = h0d
        if T > 1:
            hprev_all[1:] = out[:T - 1]
        stacked = ((x2, d1, wx1), (hprev_all.reshape(T * B, H), d1, wh1), (x2, d2, wx2), (RH.reshape(T * B, H), d2, wh2))
        for idx, (a, d, w) in enumerate(stacked, start=2):
            if not w.requires_grad:
                continue
            if _is_leaf_f32(w):
                hp.gemm(a.T, d, w.grad, beta=1.0)
            else:
                dw = hp.empty(w.shape, np.float32)
                hp.gemm(a.T, d, dw)
                grads[idx] = dw
        if self.has_bias:
            if b1.requires_grad:
                grads[6] = d1.sum(0).reshape(b1.shape)
            if b2.requires_grad:
                grads[7] = d2.sum(0).reshape(b2.shape)
        return grads
