"""Recurrent cells and sequence wrappers (surface of pydynet/nn/modules/rnn.py:13-723).

Gate algebra follows the reference, which is NOT PyTorch's: in `GRUCell` the reset gate
multiplies h BEFORE the Wh2 product and z weights the candidate (rnn.py:537-544).
The stacking / bidirectional wiring of the sequence modules is the reference's as well
(deeper layers consume the forward outputs of the layer below, rnn.py:660-694)."""
import math

import numpy as np

from .module import Module
from .. import init, functional as F
from ..parameter import Parameter
from ...special import empty, zeros
from ... import core
from ...cuda import Device


class _Cell(Module):
    _gates = 1          # width multiplier of the fused gate matrices

    def _setup(self, input_size, hidden_size, bias, device, dtype):
        self.input_size, self.hidden_size, self.has_bias = input_size, hidden_size, bias
        self.kwargs = {"device": Device(device), "dtype": dtype}

    def _uniform(self, *params):
        bound = math.sqrt(1 / self.hidden_size)
        for p in params:
            init.uniform_(p, -bound, bound)

    def init_hidden(self, x):
        assert x.ndim in {1, 2}
        shape = self.hidden_size if x.ndim == 1 else (x.shape[0], self.hidden_size)
        return zeros(shape, **self.kwargs)

    def _check(self, x, h, what="hidden"):
        assert (x.ndim == 1 and h.shape == (self.hidden_size,)) or (
            x.ndim == 2 and h.shape == (x.shape[0], self.hidden_size)), f"Wrong {what} state input!"

    def move(self, device):
        self.kwargs['device'] = device
        return super().move(device)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.input_size}, {self.hidden_size}, bias={self.has_bias})"


class RNNCell(_Cell):
    def __init__(self, input_size, hidden_size, bias=True, nonlinearity='tanh', device=None, dtype=None) -> None:
        super().__init__()
        self._setup(input_size, hidden_size, bias, device, dtype)
        self.nonlinearity = nonlinearity
        self.fn = {'tanh': F.tanh, 'relu': F.relu}[nonlinearity]
        self.Wx = Parameter(empty((input_size, hidden_size), **self.kwargs))
        self.Wh = Parameter(empty((hidden_size, hidden_size), **self.kwargs))
        if bias:
            self.bias = Parameter(empty(hidden_size, **self.kwargs))
        self.reset_paramters()

    def reset_paramters(self):
        self._uniform(self.Wx, self.Wh, *([self.bias] if self.has_bias else []))

    def forward(self, x, h=None):
        if h is None:
            h = self.init_hidden(x)
        else:
            self._check(x, h)
        if x.device.is_hip and x.ndim == 2 and h.ndim == 2 and x.dtype == np.float32 == h.dtype == self.Wx.dtype:
            from ...core.fused import rnn_cell          # one tape node: 2 GEMMs + 1 pointwise kernel
            return rnn_cell(x, h, self.Wx, self.Wh, self.bias if self.has_bias else None, self.nonlinearity)
        lin = x @ self.Wx + h @ self.Wh
        if self.has_bias:
            lin = lin + self.bias
        return self.fn(lin)


class GRUCell(_Cell):
    def __init__(self, input_size, hidden_size, bias=True, device=None, dtype=None) -> None:
        super().__init__()
        self._setup(input_size, hidden_size, bias, device, dtype)
        self.Wx1 = Parameter(empty((input_size, 2 * hidden_size), **self.kwargs))
        self.Wh1 = Parameter(empty((hidden_size, 2 * hidden_size), **self.kwargs))
        self.Wx2 = Parameter(empty((input_size, hidden_size), **self.kwargs))
        self.Wh2 = Parameter(empty((hidden_size, hidden_size), **self.kwargs))
        if bias:
            self.bias1 = Parameter(empty(2 * hidden_size, **self.kwargs))
            self.bias2 = Parameter(empty(hidden_size, **self.kwargs))
        self.reset_parameters()

    def reset_parameters(self):
        # draw order of the reference: Wx1, Wx2, Wh1, Wh2, bias1, bias2 (rnn.py:546-554)
        self._uniform(self.Wx1, self.Wx2, self.Wh1, self.Wh2,
                      *([self.bias1, self.bias2] if self.has_bias else []))

    def forward(self, x, h=None):
        if h is None:
            h = self.init_hidden(x)
        else:
            self._check(x, h)
        if x.device.is_hip and x.ndim == 2 and h.ndim == 2 and x.dtype == np.float32 == h.dtype:
            from ...core.fused import gru_cell          # one tape node: 4 GEMMs + 2 gate kernels
            return gru_cell(x, h, self.Wx1, self.Wh1, self.Wx2, self.Wh2,
                            *((self.bias1, self.bias2) if self.has_bias else ()))
        lin1 = x @ self.Wx1 + h @ self.Wh1
        if self.has_bias:
            lin1 = lin1 + self.bias1
        z, r = core.split(F.sigmoid(lin1), 2, axis=1)
        lin2 = x @ self.Wx2 + (r * h) @ self.Wh2
        if self.has_bias:
            lin2 = lin2 + self.bias2
        return (1 - z) * h + z * F.tanh(lin2)


class LSTMCell(_Cell):
    def __init__(self, input_size, hidden_size, bias=True, device=None, dtype=None) -> None:
        super().__init__()
        self._setup(input_size, hidden_size, bias, device, dtype)
        self.Wx = Parameter(empty((input_size, 4 * hidden_size), **self.kwargs))
        self.Wh = Parameter(empty((hidden_size, 4 * hidden_size), **self.kwargs))
        if bias:
            self.bias = Parameter(empty(4 * hidden_size, **self.kwargs))
        self.reset_paramters()

    def reset_paramters(self):
        self._uniform(self.Wx, self.Wh, *([self.bias] if self.has_bias else []))

    def forward(self, x, hx=None):
        if hx is None:
            h, c = self.init_hidden(x), self.init_hidden(x)
        else:
            h, c = hx
            self._check(x, h)
            self._check(x, c, "cell")
        if (x.device.is_hip and x.ndim == 2 and h.ndim == 2 and c.ndim == 2
                and x.dtype == np.float32 == h.dtype == c.dtype == self.Wx.dtype):
            from ...core.fused import lstm_cell         # one tape node; its value is the packed [h' | c']
            hc = lstm_cell(x, h, c, self.Wx, self.Wh, self.bias if self.has_bias else None)
            H = self.hidden_size
            return hc[:, :H], hc[:, H:]
        lin = x @ self.Wx + h @ self.Wh
        if self.has_bias:
            lin = lin + self.bias
        fio, g = core.hsplit(lin, [3 * self.hidden_size])
        f, i, o = core.hsplit(F.sigmoid(fio), 3)
        c = f * c + i * F.tanh(g)
        return o * F.tanh(c), c


class _Recurrent(Module):
    """Shared sequence driver: python loop over time, outputs stacked with concat."""
    _cell_cls = None
    _prefix = "rnn"
    _has_cell_state = False

    def _build(self, input_size, hidden_size, num_layers, bias, batch_first, bidirectional, device, dtype, **cell_kw):
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.has_bias, self.batch_first, self.bidirectional = bias, batch_first, bidirectional
        self.kwargs = {"device": Device(device), "dtype": dtype}
        assert num_layers > 0
        sizes = [input_size] + [hidden_size] * (num_layers - 1)
        self.cells, self.rcells = [], []
        for i in range(num_layers):
            cell = self._cell_cls(sizes[i], hidden_size, bias, **cell_kw, **self.kwargs)
            setattr(self, f'{self._prefix}_{i}', cell)
            self.cells.append(cell)
        if bidirectional:
            for i in range(num_layers):
                cell = self._cell_cls(sizes[i], hidden_size, bias, **cell_kw, **self.kwargs)
                setattr(self, f'r{self._prefix}_{i}', cell)
                self.rcells.append(cell)

    def init_hidden(self, x):
        assert x.ndim in {2, 3}
        d = 2 if self.bidirectional else 1
        shape = (d * self.num_layers, self.hidden_size) if x.ndim == 2 else \
            (d * self.num_layers, x.shape[1], self.hidden_size)
        return zeros(shape, **self.kwargs)

    def cell_forward(self, cell, x, h, c=None):
        outs, cs = [], []
        for t in range(x.shape[0]):
            if self._has_cell_state:
                h, c = cell(x[t], (h, c))
            else:
                h = cell(x[t], h)
            outs.append(core.unsqueeze(h, axis=0))
        return (outs, h, c)

    def _run(self, x, h, c=None):
        L, bi = self.num_layers, self.bidirectional
        hn, cn, rhn, rcn = [], [], [], []
        f_list = r_list = None
        for i in range(L):
            f_in = x if i == 0 else core.concat(f_list)
            f_list, _, c_last = self.cell_forward(self.cells[i], f_in, h[i], None if c is None else c[i])
            hn.append(f_list[-1]); cn.append(c_last)
            if bi:
                r_in = x[::-1] if i == 0 else core.concat(r_list)
                r_list, _, rc_last = self.cell_forward(self.rcells[i], r_in, h[i + L], None if c is None else c[i + L])
                rhn.append(r_list[-1]); rcn.append(rc_last)
        if bi:
            output = core.concat([core.concat(f_list), core.concat(r_list[::-1])], axis=-1)
        else:
            output = core.concat(f_list)
        h_all = hn + rhn
        h_out = h_all[0] if len(h_all) == 1 else core.concat(h_all)
        c_out = None
        if c is not None:
            c_all = [core.unsqueeze(t, 0) for t in cn + rcn]
            c_out = c_all[0] if len(c_all) == 1 else core.concat(c_all)
        return output, h_out, c_out

    def move(self, device):
        self.kwargs['device'] = device
        return super().move(device)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}({self.input_size}, {self.hidden_size}, num_layers={self.num_layers}, "
                f"bias={self.has_bias}, batch_first={self.batch_first}, bidirectional={self.bidirectional})")


class _PlainRecurrent(_Recurrent):
    def forward(self, x, h=None):
        if self.batch_first and x.ndim == 3:
            x = x.swapaxes(0, 1)
        if h is None:
            h = self.init_hidden(x)
        else:
            d = 2 if self.bidirectional else 1
            assert (x.ndim == 2 and h.shape == (d * self.num_layers, self.hidden_size)) or (
                x.ndim == 3 and h.shape == (d * self.num_layers, x.shape[1], self.hidden_size)), \
                "Wrong hidden state input!"
        output, hn, _ = self._run(x, h)
        if self.batch_first and x.ndim == 3:
            output, hn = output.swapaxes(0, 1), hn.swapaxes(0, 1)
        return output, hn


class RNN(_PlainRecurrent):
    _cell_cls, _prefix = RNNCell, "rnn"

    def __init__(self, input_size, hidden_size, num_layers=1, nonlinearity='tanh', bias=True,
                 batch_first=False, bidirectional=False, device=None, dtype=None) -> None:
        super().__init__()
        self.nonlinearity = nonlinearity
        self._build(input_size, hidden_size, num_layers, bias, batch_first, bidirectional, device, dtype,
                    nonlinearity=nonlinearity)
        self.RNNCells, self.rRNNCells = self.cells, self.rcells


class GRU(_PlainRecurrent):
    _cell_cls, _prefix = GRUCell, "gru"

    def forward(self, x, h=None):
        # single layer, one direction, batched sequence on the HIP device: the whole time loop is one
        # fused node (hoisted input projections, 4 launches per step forward, 5 backward)
        if (self.num_layers == 1 and not self.bidirectional and x.ndim == 3 and x.device.is_hip
                and x.dtype == np.float32 and self.cells[0].Wx1.dtype == np.float32):
            from ...core.fused import gru_sequence
            xs = x.swapaxes(0, 1) if self.batch_first else x
            h0 = self.init_hidden(xs) if h is None else h
            assert h0.shape == (1, xs.shape[1], self.hidden_size), "Wrong hidden state input!"
            c = self.cells[0]
            out = gru_sequence(xs, h0[0], c.Wx1, c.Wh1, c.Wx2, c.Wh2, *((c.bias1, c.bias2) if c.has_bias else ()))
            hn = out[xs.shape[0] - 1:]
            if self.batch_first:
                out, hn = out.swapaxes(0, 1), hn.swapaxes(0, 1)
            return out, hn
        return super().forward(x, h)

    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, batch_first=False,
                 bidirectional=False, device=None, dtype=None) -> None:
        super().__init__()
        self._build(input_size, hidden_size, num_layers, bias, batch_first, bidirectional, device, dtype)
        self.GRUCells, self.rGRUCells = self.cells, self.rcells


class LSTM(_Recurrent):
    _cell_cls, _prefix, _has_cell_state = LSTMCell, "lstm", True

    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, batch_first=False,
                 bidirectional=False, device=None, dtype=None) -> None:
        super().__init__()
        self._build(input_size, hidden_size, num_layers, bias, batch_first, bidirectional, device, dtype)
        self.LSTMCells, self.rLSTMCells = self.cells, self.rcells

    def forward(self, x, hx=None):
        if self.batch_first and x.ndim == 3:
            x = x.swapaxes(0, 1)
        h, c = (self.init_hidden(x), self.init_hidden(x)) if hx is None else hx
        output, hn, cn = self._run(x, h, c)
        if self.batch_first and x.ndim == 3:
            output, hn, cn = output.swapaxes(0, 1), hn.swapaxes(0, 1), cn.swapaxes(0, 1)
        return output, (hn, cn)
