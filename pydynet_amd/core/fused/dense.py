"""Projection, embedding and loss nodes: linear, embedding, cross_entropy, linear_cross_entropy (lm_head + loss as one node).
(One module of `pydynet_amd.core.fused`; the package docstring lists the reference chains each node replaces.)"""
from __future__ import annotations

import contextlib
import os

import numpy as np

from ...autograd import is_grad_enable
from ..tensor import Tensor, _Operator
from ._common import _hip, _L, _contig, _require_f32, _foldable, _beside, _is_leaf_f32, _Deferred, hip_f32


class linear(_Deferred, _Operator):
    """y = x @ W (+ b) (+ residual) over the last axis of x; W is (in, out).

    `residual` (shape of y) folds the `z = x + sublayer(x)` add of a transformer block into the
    GEMM epilogue; its gradient is the upstream gradient itself.

    A projection `relu` could take over (float32 on the device, out features a multiple of 32, no residual) is
    created without running anything (`_Deferred`): `F.relu` of it becomes ONE `linear_relu` node, any other consumer
    reads `.data`, which runs the product then."""

    folds_existing = True      # backward adds the gradient x already holds inside the dX GEMM
    defer = True               # class switch: False runs every product at construction (no linear + relu fusion)
    _reshape_hook = True       # Tensor.reshape asks core/fused/chain.py about pending projections

    def __init__(self, x, weight, bias=None, residual=None):
        self.has_bias, self.has_res = bias is not None, residual is not None
        ins = [x, weight] + ([bias] if self.has_bias else []) + ([residual] if self.has_res else [])
        if type(self) is linear and self.defer and linear_relu.applicable(x, weight, bias, residual):
            self._init_deferred(ins, tuple(x.shape[:-1]) + (weight.shape[1],), np.float32)
        else:
            super().__init__(*ins)

    def _split(self, ins):
        b = ins[2] if self.has_bias else None
        r = ins[2 + self.has_bias] if self.has_res else None
        return ins[0], ins[1], b, r

    def forward_(self, *ins):
        x, w, b, r = self._split(ins)
        if self.xp is np:
            y = x.data @ w.data
            if b is not None:
                y = y + b.data
            return y + r.data if r is not None else y
        _require_f32(self, x, w, b, r)
        hp = _hip()
        fin, fout = w.shape
        x2 = x.data.reshape(-1, fin)
        out = hp.empty(x.shape[:-1] + (fout,), np.float32)
        res = _contig(r.data).reshape(-1, fout) if r is not None else None
        hp.gemm(x2, w.data, out.reshape(-1, fout), bias=b.data.reshape(-1) if b is not None else None,
                residual=res)
        return out

    def _dx(self, hp, g2, x, w, fin):
        dx = hp.empty(x.shape, np.float32)
        ex = _foldable(self, 0, x)
        mask = getattr(x, "_relu_bits", None) if type(x) is linear_relu else None
        if mask is not None and g2._strides[-1] == 1 and (ex is None or ex.is_contiguous()):
            # x = relu(pre-activation) of a linear_relu node: its bits applied in this product's store, so that node
            # receives the gradient of its PRE-activation (relu'(z) o (g W^T + what x already holds)) -- no relu pass
            wd = w.data
            # ... and the column sums of that gradient (the bias gradient of the layer below) leave the same store as one
            # partial row per 32 rows: no pass over dx for them
            xb = x.last[2] if (x.has_bias and len(x.last) > 2) else None
            parts = hp.empty(((g2.shape[0] + 31) // 32, fin), np.float32) if (xb is not None and xb.requires_grad) else None
            _L().call("pdn_linear_dx_masked_f32", g2._ptr, g2._strides[0], wd._ptr, wd._strides[0], wd._strides[1],
                      dx._ptr, fin, ex._ptr if ex is not None else None, mask._ptr,
                      parts._ptr if parts is not None else None, g2.shape[0], fin, wd.shape[1], hp.stream())
            dx._aux = ("relu_masked", mask, parts.sum(0) if parts is not None else None)
            return dx
        hp.gemm(g2, w.data.T, dx.reshape(-1, fin),                         # NT
                residual=ex.reshape(-1, fin) if ex is not None else None)
        return dx

    def backward_all(self, g):
        x, w, b, r = self._split(self.last)
        fin, fout = w.shape
        grads = [None] * len(self.last)
        if self.has_res and r.requires_grad:
            grads[2 + self.has_bias] = g
        if self.xp is np:
            g2, x2 = g.reshape(-1, fout), x.data.reshape(-1, fin)
            if x.requires_grad:
                grads[0] = (g2 @ w.data.T).reshape(x.shape)
            if w.requires_grad:
                grads[1] = x2.T @ g2
            if b is not None and b.requires_grad:
                grads[2] = g2.sum(0).reshape(b.shape)
            return grads
        hp = _hip()
        g2 = _contig(g).reshape(-1, fout)
        x2 = x.data.reshape(-1, fin)
        side = _beside(hp, fin, fout, x.requires_grad and w.requires_grad)
        if x.requires_grad and side is None:
            grads[0] = self._dx(hp, g2, x, w, fin)
        need_db = b is not None and b.requires_grad
        aux = getattr(g, "_aux", None)
        if need_db and aux is not None and aux[0] == "colsum" and aux[1].size == b.size:
            # the producer of g (fused cross entropy) already summed its columns
            if _is_leaf_f32(b):
                b.grad += aux[1].reshape(b.grad.shape)
            else:
                grads[2] = aux[1].reshape(b.shape)
            need_db = False
        # bias gradient = column sums of g: formed inside the dW GEMM (both read g once) when the
        # operands have the aligned x^T @ g layout and the leaf buffers can be accumulated into
        # (fusing costs ~25 % of the dW GEMM, a separate pass over g one read of it: the fusion only
        # pays for short contractions, fin < ~8 * MFMA rate / HBM rate ~ 192)
        fuse_db = (need_db and w.requires_grad and _is_leaf_f32(b) and x2.is_contiguous() and fin < 192
                   and fin % 4 == 0 and fout % 4 == 0 and x2.shape[0] % 4 == 0)
        if w.requires_grad:
            cs = b.grad.reshape(-1) if fuse_db else None
            dw = None if _is_leaf_f32(w) else hp.empty((fin, fout), np.float32)
            with side or contextlib.nullcontext():
                if dw is None:
                    hp.gemm(x2.T, g2, w.grad, beta=1.0, b_colsum=cs, colsum_accumulate=True)   # TN, += into the leaf
                else:
                    hp.gemm(x2.T, g2, dw, b_colsum=cs, colsum_accumulate=True)
                    grads[1] = dw
        if side is not None:
            grads[0] = self._dx(hp, g2, x, w, fin)
            side.join()
        if need_db and not fuse_db:
            grads[2] = g2.sum(0).reshape(b.shape)
        return grads


class linear_relu(linear):
    """relu(x @ W + b) as ONE node (examples/pydynet/mnist.py:70-78: Linear -> ReLU; nn/functional.py:31-32 relu =
    maximum(0., x), tensor.py:808-814: its gradient passes where out == x, i.e. pre-activation >= 0).

    Forward: the product stores max(0, x W + b) and one bit per element (pre-activation >= 0) -- the pre-activation
    never exists in HBM (`pdn_linear_relu_fwd_f32`).  Backward: a `linear` consumer applies the bits inside its
    input-gradient product (`linear._dx`), so the gradient arrives as that of the pre-activation; a gradient from any
    other consumer (or an accumulated one) gets the bits applied here (`pdn_relu_mask_bwd_f32`; applying them twice
    changes nothing)."""

    # below this many rows the plain product wins: it may split the contraction over workgroups to fill the chip (a 256-row
    # batch is 16 tiles), which the stores with bits cannot (measured: the 256-sample MLP step 0.205 -> 0.220 ms when fused)
    min_rows = 4096

    @staticmethod
    def applicable(x, weight, bias, residual=None):
        if (residual is not None or x.ndim < 1 or weight.ndim != 2 or not x.device.is_hip
                or not hip_f32(x, weight, bias)):
            return False
        rows = int(np.prod(x.shape[:-1], dtype=np.int64))
        return (rows >= linear_relu.min_rows and weight.shape[1] % 32 == 0 and x.shape[-1] == weight.shape[0]
                and (bias is None or bias.size == weight.shape[1]))

    def forward_(self, *ins):
        x, w, b, _ = self._split(ins)
        hp, L = _hip(), _L()
        fin, fout = w.shape
        x2 = _contig(x.data).reshape(-1, fin)
        out = hp.empty(x.shape[:-1] + (fout,), np.float32)
        self._relu_bits = hp.empty((x2.shape[0], fout // 32), np.float32)      # opaque 32-bit words
        wd = w.data
        L.call("pdn_linear_relu_fwd_f32", x2._ptr, x2._strides[0], wd._ptr, wd._strides[0], wd._strides[1],
               _contig(b.data).reshape(-1)._ptr if b is not None else None, out._ptr, fout, self._relu_bits._ptr,
               x2.shape[0], fout, fin, hp.stream())
        return out

    def backward_all(self, g):
        aux = getattr(g, "_aux", None)
        if not (aux is not None and aux[0] == "relu_masked" and aux[1] is self._relu_bits):
            hp = _hip()
            g = _contig(g)
            dz = hp.empty(g.shape, np.float32)
            _L().call("pdn_relu_mask_bwd_f32", g._ptr, self._relu_bits._ptr, dz._ptr, g.size // g.shape[-1], g.shape[-1],
                      hp.stream())
            g = dz
        elif len(aux) > 2 and aux[2] is not None:
            g._aux = ("colsum", aux[2])          # the producer summed the columns of this very array (see linear._dx)
        return super().backward_all(g)


class embedding(_Operator):
    """out = W[ids]; gradient = scatter-ASSIGN of the last occurrence of each id (reference
    semantics, tensor.py:937-940); `accumulate=True` opts into torch-style scatter-add."""

    accumulate = False

    def __init__(self, ids, weight):
        if isinstance(ids, Tensor):
            ids = ids.data
        self._ids = ids
        super().__init__(weight)

    def forward_(self, w):
        if self.xp is np:
            return w.data[self._ids]
        hp = _hip()
        if isinstance(self._ids, np.ndarray) or not hasattr(self._ids, "_ptr"):
            self._ids = hp.from_numpy(np.asarray(self._ids).astype(np.int64))
        return w.data[self._ids]

    def backward_all(self, g):
        w = self.last[0]
        # data parallel (distributed.DataParallel): per-row owner rank, so that the summed gradient keeps
        # the scatter-ASSIGN semantics of the concatenated batch; irrelevant for scatter-add
        owner = tag = None
        if getattr(w, "_dp_owner", None) is not None and not self.accumulate:
            owner, tag = w._dp_owner(self._ids, w.shape[0])
        if self.xp is np:
            full = np.zeros(w.shape, dtype=w.dtype)
            if self.accumulate:
                np.add.at(full, self._ids, g)
            elif owner is not None:
                ids = np.asarray(self._ids).reshape(-1)
                keep = owner[ids] == tag
                full[ids[keep]] = g.reshape(ids.size, -1)[keep]
            else:
                full[self._ids] = g
            return [full]
        hp, L = _hip(), _L()
        V, D = w.shape
        g = _contig(g)
        ids = _contig(self._ids)
        ws, wsb = hp.workspace(V * 4)
        optr, otag = (owner._ptr, tag) if owner is not None else (None, 0.0)
        if _is_leaf_f32(w):
            L.call("pdn_embedding_scatter_f32", g._ptr, ids._ptr, ids.size, w.grad._ptr, V, D,
                   2 if self.accumulate else 1, optr, otag, ws, wsb, hp.stream())
            return [None]
        full = hp.zeros(w.shape, np.float32)
        L.call("pdn_embedding_scatter_f32", g._ptr, ids._ptr, ids.size, full._ptr, V, D,
               2 if self.accumulate else 0, optr, otag, ws, wsb, hp.stream())
        return [full]


class cross_entropy(_Operator):
    """mean / sum over rows of  logsumexp(x_n) - x_n[t_n]   (integer targets)."""

    def __init__(self, logits, targets, reduction="mean"):
        if reduction not in ("mean", "sum"):
            raise ValueError("reduction must be mean or sum.")
        self.reduction = reduction
        self._t = targets.data if isinstance(targets, Tensor) else targets
        super().__init__(logits)

    def forward_(self, x):
        n, V = x.shape
        if self.xp is np:
            t = np.asarray(self._t)
            m = x.data.max(-1, keepdims=True)
            self._lse = np.log(np.exp(x.data - m).sum(-1, keepdims=True)) + m
            rows = self._lse[:, 0] - x.data[np.arange(n), t]
            return rows.mean() if self.reduction == "mean" else rows.sum()
        _require_f32(self, x)
        hp, L = _hip(), _L()
        if not hasattr(self._t, "_ptr"):
            self._t = hp.from_numpy(np.asarray(self._t).astype(np.int64))
        self._x = _contig(x.data)
        self._t = _contig(self._t)
        loss_row = hp.empty((n,), np.float32)
        self._lse = hp.empty((n,), np.float32)
        out = hp.empty((1,), np.float32)
        mean = 1 if self.reduction == "mean" else 0
        self._dx = None
        from ...autograd import is_grad_enable
        if x.requires_grad and is_grad_enable():
            # the gradient w.r.t. the logits needs nothing computed later: write it now, while each
            # row is still in L2 (one pass over HBM for forward + backward)
            self._dx = hp.empty((n, V), np.float32)
            # column sums of dlogits (= the bias gradient of the Linear that made the logits) come
            # for free while the rows stream through; handed to that Linear via the array's `_aux`
            wsb = L.query("pdn_cross_entropy_colsum_workspace_bytes", n, V)
            cs = hp.empty((V,), np.float32) if wsb else None
            ws, wsb = hp.workspace(wsb) if wsb else (None, 0)
            L.call("pdn_cross_entropy_fwd_bwd_f32", self._x._ptr, self._t._ptr, n, V, mean,
                   1.0 / n if mean else 1.0, loss_row._ptr, self._lse._ptr, out._ptr, self._dx._ptr,
                   cs._ptr if cs is not None else None, ws, wsb, hp.err_flag_ptr(), hp.stream())
            self._dx._aux = ("colsum", cs) if cs is not None else None
        else:
            L.call("pdn_cross_entropy_fwd_f32", self._x._ptr, self._t._ptr, n, V, mean, loss_row._ptr,
                   self._lse._ptr, out._ptr, hp.err_flag_ptr(), hp.stream())
        return out.reshape(())

    def backward_all(self, g):
        x = self.last[0]
        n, V = x.shape
        scale = 1.0 / n if self.reduction == "mean" else 1.0
        if self.xp is np:
            sm = np.exp(x.data - self._lse)
            sm[np.arange(n), np.asarray(self._t)] -= 1
            return [sm * (g * np.asarray(scale, x.dtype))]
        hp, L = _hip(), _L()
        g = _contig(g)
        if self._dx is not None:
            dx, self._dx = self._dx, None       # written in forward; apply the upstream scalar (1 -> no-op)
            L.call("pdn_scale_by_device_scalar_f32", dx._ptr, dx.size, g._ptr, hp.stream())
            aux = getattr(dx, "_aux", None)
            if aux is not None:
                L.call("pdn_scale_by_device_scalar_f32", aux[1]._ptr, aux[1].size, g._ptr, hp.stream())
            return [dx]
        dx = hp.empty((n, V), np.float32)
        L.call("pdn_cross_entropy_bwd_f32", self._x._ptr, self._t._ptr, self._lse._ptr, g._ptr,
               scale, dx._ptr, n, V, hp.stream())
        return [dx]


class linear_cross_entropy(_Operator):
    """loss = cross_entropy(x @ W + b, targets) as ONE tape node (llm/llama/model.py:179 feeding
    nn/functional.py:364-381): the (rows, V) gradient of the logits never exists in memory.  Forward: the
    logits GEMM and one read-only pass for the row statistics (log-sum-exp, loss).  Backward: the two products
    `dlogits @ W^T` and `x^T @ dlogits` (and the column sums for the bias) form
    dlogits = (softmax - onehot) * scale from the saved logits as they consume it (`pdn_linear_ce_backward_f32`).
    Versus linear + cross_entropy nodes: one (rows x V) write and none of its re-reads less.
    Deferred form (training step): the input-gradient product runs in the FORWARD pass (it is where the sum of exponentials
    comes from), so (i) a forward under grad mode that is never followed by `backward()` still pays that product -- wrap
    validation losses in `no_grad()` -- and (ii) dx is formed from the weights as they are at forward time: updating the
    weight IN PLACE between forward and backward leaves dx consistent with the forward pass (as the reference's tape is)
    but not with a dW computed from the new weights; `backward_all` asserts that the weight buffer is still the one
    forward read."""

    folds_existing = True
    enabled = True
    min_rows = int(os.environ.get("PDN_LINCE_MIN_ROWS", "16384"))
    lse_epilogue = os.environ.get("PDN_NO_LSE_EPILOGUE", "0") != "1"      # row statistics in the projection's store (A/B switch)
    # statistics split over the projection (row maxima) and the input-gradient product (sum of exponentials), the latter
    # run in the forward pass of a training step (A/B switch)
    deferred_norm = os.environ.get("PDN_NO_CE_DEFERRED", "0") != "1"

    @staticmethod
    def applicable(x, w, b, targets, reduction="mean"):
        if not (linear_cross_entropy.enabled and x.device.is_hip and x.dtype == np.float32 and w.dtype == np.float32
                and (b is None or b.dtype == np.float32) and reduction in ("mean", "sum") and w.ndim == 2):
            return False
        rows = 1
        for d in x.shape[:-1]:
            rows *= d
        t = targets.data if isinstance(targets, Tensor) else targets
        # (below 28672 tokens the row workgroups of the output-resident input-gradient kernel no longer fill the chip:
        #  it then cuts K = vocabulary into ranges over the grid, pdn_gemm_outres_plan; below `min_rows` the separate
        #  nodes on the tiled kernels are left in place)
        return (x.shape[-1] == w.shape[0] and getattr(t, "ndim", 0) == 1 and t.shape[0] == rows
                and rows >= linear_cross_entropy.min_rows
                and bool(_L().query("pdn_linear_ce_supported", rows, w.shape[1], w.shape[0])))

    def __init__(self, x, weight, bias, targets, reduction="mean"):
        self.reduction = reduction
        self.has_bias = bias is not None
        self._t = targets.data if isinstance(targets, Tensor) else targets
        super().__init__(*([x, weight] + ([bias] if self.has_bias else [])))

    def forward_(self, x, w, b=None):
        _require_f32(self, x, w, b)
        hp, L = _hip(), _L()
        fin, V = w.shape
        x2 = _contig(x.data).reshape(-1, fin)
        n = x2.shape[0]
        if not hasattr(self._t, "_ptr"):
            self._t = hp.from_numpy(np.asarray(self._t).astype(np.int64))
        self._t = _contig(self._t)
        logits = hp.empty((n, V), np.float32)
        loss_row, lse, out = hp.empty((n,), np.float32), hp.empty((n,), np.float32), hp.empty((1,), np.float32)
        wd = w.data
        in_gemm = bool(wd.is_contiguous() and x2._strides[1] == 1 and L.query("pdn_linear_lse_supported", n, V, fin))
        self.deferred = bool(linear_cross_entropy.deferred_norm and wd.is_contiguous() and x2._strides[1] == 1
                             and is_grad_enable() and x.requires_grad
                             and L.query("pdn_linear_rowmax_supported", n, V, fin)
                             and L.query("pdn_linear_ce_dx_deferred_supported", n, V, fin))
        self.stats_in_gemm = bool(linear_cross_entropy.lse_epilogue and in_gemm and not self.deferred)
        self._dxu = None
        bp = b.data._ptr if b is not None else None
        mean = 1 if self.reduction == "mean" else 0
        if self.deferred:
            # the projection leaves the row maxima; the input-gradient product -- it needs exp(logit - max) anyway, and not
            # the upstream gradient, a scalar applied in backward -- sums the exponentials as it multiplies: it runs HERE,
            # the loss follows from its log-sum-exp with one gather per row, and no pass over the logits exists
            # (few rows: both products cut the vocabulary into ranges over the grid -- `parts` vectors of maxima, a
            #  workspace of unnormalised rows and row sums)
            parts = L.query("pdn_linear_rowmax_parts", n, V, fin)
            rowmax = hp.empty((parts * n,), np.float32)
            L.call("pdn_linear_rowmax_fwd_f32", x2._ptr, wd._ptr, bp, logits._ptr, rowmax._ptr, n, V, fin, x2._strides[0],
                   V, V, hp.stream())
            self._dxu = hp.empty((n, fin), np.float32)
            self._w_ptr = wd._ptr                          # (backward checks that the weight was not re-homed meanwhile)
            ws, wsb = hp.workspace(L.query("pdn_linear_ce_dx_deferred_workspace_bytes", n, V, fin))
            L.call("pdn_linear_ce_dx_deferred_f32", logits._ptr, rowmax._ptr, parts, self._t._ptr,
                   1.0 / n if mean else 1.0, wd._ptr, self._dxu._ptr, lse._ptr, n, V, fin, ws, wsb, hp.stream())
            L.call("pdn_cross_entropy_from_lse_f32", logits._ptr, V, lse._ptr, self._t._ptr, n, V, mean, loss_row._ptr,
                   out._ptr, hp.err_flag_ptr(), hp.stream())
        elif self.stats_in_gemm:
            # the projection leaves the rows' log-sum-exp itself (transposed accumulators: a lane owns a token): no
            # pass over the logits for the statistics, the loss is one gather per row
            L.call("pdn_linear_lse_fwd_f32", x2._ptr, wd._ptr, bp, logits._ptr, lse._ptr, n, V, fin, x2._strides[0], V, V,
                   hp.stream())
            L.call("pdn_cross_entropy_from_lse_f32", logits._ptr, V, lse._ptr, self._t._ptr, n, V, mean, loss_row._ptr,
                   out._ptr, hp.err_flag_ptr(), hp.stream())
        else:
            hp.gemm(x2, wd, logits, bias=b.data.reshape(-1) if b is not None else None)
            L.call("pdn_cross_entropy_fwd_f32", logits._ptr, self._t._ptr, n, V, mean, loss_row._ptr, lse._ptr, out._ptr,
                   hp.err_flag_ptr(), hp.stream())
        self._saved = (x2, logits, lse)
        return out.reshape(())

    def backward_all(self, g):
        hp, L = _hip(), _L()
        x, w = self.last[0], self.last[1]
        b = self.last[2] if self.has_bias else None
        fin, V = w.shape
        x2, logits, lse = self._saved
        self._saved = None
        n = x2.shape[0]
        g = _contig(g)
        grads = [None] * len(self.last)
        dx = ex = None
        dxu, self._dxu = self._dxu, None
        if dxu is not None and getattr(self, "_w_ptr", None) not in (None, w.data._ptr):
            raise RuntimeError("linear_cross_entropy: the weight's buffer was replaced between forward and backward; the "
                               "input gradient of the deferred form was formed from the forward pass's weights")
        if x.requires_grad and dxu is not None:
            dxu *= g.reshape(())                           # formed in the forward pass, up to the upstream scalar
            grads[0] = dxu.reshape(x.shape)
        elif x.requires_grad:
            dx = hp.empty(x.shape, np.float32)
            ex = _foldable(self, 0, x)
            grads[0] = dx
        dw, dw_beta = None, 0.0
        if w.requires_grad:
            if _is_leaf_f32(w):
                dw, dw_beta = w.grad, 1.0
            else:
                dw = grads[1] = hp.empty((fin, V), np.float32)
        db, db_beta = None, 0.0
        if b is not None and b.requires_grad:
            if _is_leaf_f32(b):
                db, db_beta = b.grad.reshape(-1), 1.0
            else:
                db = hp.empty((V,), np.float32)
                grads[2] = db.reshape(b.shape)
        ws, wsb = hp.workspace(L.query("pdn_linear_ce_workspace_bytes", n, V, fin)) if (dw is not None or db is not None) else (None, 0)
        L.call("pdn_linear_ce_backward_f32", x2._ptr, x2._strides[0], logits._ptr, lse._ptr, self._t._ptr,
               1.0 / n if self.reduction == "mean" else 1.0, g._ptr, w.data._ptr,
               dx._ptr if dx is not None else None, ex._ptr if ex is not None else None,
               dw._ptr if dw is not None else None, dw_beta, db._ptr if db is not None else None, db_beta,
               n, V, fin, ws, wsb, hp.stream())
        return grads
