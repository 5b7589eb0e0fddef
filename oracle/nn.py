"""ORACLE (test infrastructure): NumPy restatement of the reference's nn / functional / optim
layer for the hot path, built on `oracle.tape`.  Citations are into /root/reference/pydynet/.

Layers are plain functions over `Var` parameters held in ordinary dicts (no Module class):
the names used as dict keys are the reference's registered parameter names, so golden
fixtures address parameters identically on both sides.
"""
from __future__ import annotations

import math

import numpy as np

from . import tape as T
from .tape import Var


# ---- parameters and initialisers (nn/parameter.py:4-15, nn/init.py:19-92) --------------------
def param(value, requires_grad=True):
    value = np.asarray(value)
    return Var(value, dtype=value.dtype, requires_grad=requires_grad)


def _fan(shape):
    fan_in, fan_out = shape[:2]
    if len(shape) > 2:
        r = math.prod(shape[2:])
        fan_in *= r
        fan_out *= r
    return fan_in, fan_out


def uniform_(shape, a, b, dtype):
    out = np.empty(shape, dtype=dtype)
    out[...] = np.random.uniform(a, b, shape)   # init.py:29-33 (float64 draw, cast on store)
    return out


def kaiming_uniform(shape, dtype, a=math.sqrt(5)):
    fan_in, _ = _fan(shape)
    gain = math.sqrt(2.0 / (1 + a ** 2))        # init.py:15-16 'leaky_relu'... see note
    return uniform_(shape, -gain * math.sqrt(3.0 / fan_in), gain * math.sqrt(3.0 / fan_in), dtype)


def kaiming_uniform_ref(shape, dtype, a=math.sqrt(5)):
    """init.kaiming_uniform_(w, a=sqrt(5)) with the default nonlinearity='relu' (init.py:67-78):
    calculate_gain('relu', a) ignores `a` and returns sqrt(2)."""
    fan_in, _ = _fan(shape)
    bound = math.sqrt(2.0) * math.sqrt(3.0 / fan_in)
    return uniform_(shape, -bound, bound, dtype)


def linear_params(fin, fout, bias=True, dtype=np.float32):
    """nn/modules/linear.py:12-37: weight (in,out) then bias, both drawn from host NumPy RNG."""
    p = {"weight": param(kaiming_uniform_ref((fin, fout), dtype))}
    if bias:
        bound = 1 / math.sqrt(fin) if fin > 0 else 0      # fan_in = weight.shape[0]
        p["bias"] = param(uniform_(fout, -bound, bound, dtype))
    return p


def conv2d_params(cin, cout, k, bias=True, dtype=np.float32):
    """nn/modules/conv.py:64-98"""
    shape = (cout, cin, k, k)
    p = {"weight": param(kaiming_uniform_ref(shape, dtype))}
    if bias:
        fan_in, _ = _fan(shape)       # = cout * k*k  (fan_in is shape[0] * receptive field)
        bound = 1 / math.sqrt(fan_in)
        p["bias"] = param(uniform_((1, cout, 1, 1), -bound, bound, dtype))
    return p


def gru_cell_params(fin, h, bias=True, dtype=np.float32):
    """nn/modules/rnn.py:500-554 (draw order: Wx1, Wx2, Wh1, Wh2, bias1, bias2)"""
    b = math.sqrt(1 / h)
    p = {}
    p["Wx1"] = param(uniform_((fin, 2 * h), -b, b, dtype))
    p["Wx2"] = param(uniform_((fin, h), -b, b, dtype))
    p["Wh1"] = param(uniform_((h, 2 * h), -b, b, dtype))
    p["Wh2"] = param(uniform_((h, h), -b, b, dtype))
    if bias:
        p["bias1"] = param(uniform_(2 * h, -b, b, dtype))
        p["bias2"] = param(uniform_(h, -b, b, dtype))
    return p


def rnn_cell_params(fin, h, bias=True, dtype=np.float32):
    """nn/modules/rnn.py:13-56"""
    b = math.sqrt(1 / h)
    p = {"Wx": param(uniform_((fin, h), -b, b, dtype)), "Wh": param(uniform_((h, h), -b, b, dtype))}
    if bias:
        p["bias"] = param(uniform_(h, -b, b, dtype))
    return p


# ---- functional (nn/functional.py) ----------------------------------------------------------
def linear(x, w, b=None):                       # :7-11
    y = x @ w
    return y + b if b is not None else y


def embedding(ids, w, padding_idx=None):        # :14-20
    q = w[ids]
    if padding_idx is not None:
        with T.no_grad():
            mask = T.unsqueeze(Var(ids.value != padding_idx if isinstance(ids, Var) else ids != padding_idx), -1)
        q = q * mask
    return q


def relu(x): return T.maximum(0.0, x)                         # :31-32
def leaky_relu(x, alpha): return T.maximum(x, alpha * x)      # :35-36
def silu(x): return x / (1 + T.exp(-x))                       # :39-40
sigmoid, tanh = T.sigmoid, T.tanh


def softmax(x, axis=None):                      # :43-49
    with T.no_grad():
        m = x.max(axis, keepdims=True)
    e = T.exp(x - m)
    return e / T.sum(e, axis=axis, keepdims=True)


def log_softmax(x, axis=None, keepdims=False):  # :52-58
    with T.no_grad():
        m = x.max(axis, keepdims=True)
    s = x - m
    return s - T.log(T.sum(T.exp(s), axis=axis, keepdims=keepdims))


def pad2d(x, p):                                # :235-251
    x = x if isinstance(x, Var) else Var(x)
    out = np.pad(x.value, [(0, 0), (0, 0), (p, p), (p, p)], "constant")
    return T._node(out, [(x, (lambda g: g[...]) if p == 0 else (lambda g: g[..., p:-p, p:-p]))])


def im2col2d(x, k, stride):                     # :194-232  col layout (N, C, kh, kw, oh, ow)
    n, c, h, w = x.shape
    oh, ow = (h - k) // stride + 1, (w - k) // stride + 1
    s0, s1, s2, s3 = x.value.strides
    strides = (s0, s1, s2, s3, s2 * stride, s3 * stride)
    shape = (n, c, k, k, oh, ow)
    col = np.lib.stride_tricks.as_strided(x.value, shape=shape, strides=strides).copy()

    def vjp(g):
        gx = np.zeros(x.shape, dtype=col.dtype)
        view = np.lib.stride_tricks.as_strided(gx, shape=shape, strides=strides)
        np.add.at(view, (...,), g)              # col2im scatter-add on the overlapping view
        return gx

    return T._node(col, [(x, vjp)])


def conv2d(x, kernel, padding=0, stride=1):     # :254-281
    n = x.shape[0]
    o, _, k, _ = kernel.shape
    col = im2col2d(pad2d(x, padding), k, stride)
    oh, ow = col.shape[-2:]
    col = col.transpose(0, 4, 5, 1, 2, 3).reshape(n * oh * ow, -1)
    out = col @ kernel.reshape(o, -1).transpose()
    return out.reshape(n, oh, ow, -1).transpose(0, 3, 1, 2)


def _pool2d(x, k, stride, padding, fn):          # :284-339
    n, c = x.shape[:2]
    col = im2col2d(pad2d(x, padding), k, stride)
    oh, ow = col.shape[-2:]
    col = col.transpose(0, 4, 5, 1, 2, 3).reshape(-1, k * k)
    out = fn(col)
    return out.reshape(n, oh, ow, c).transpose(0, 3, 1, 2)


def max_pool2d(x, k, stride, padding=0): return _pool2d(x, k, stride, padding, lambda c: c.max(1))
def avg_pool2d(x, k, stride, padding=0): return _pool2d(x, k, stride, padding, lambda c: c.mean(1))


def mse_loss(pred, true, reduction="mean"):     # :342-350
    s = T.square(pred - true)
    return T.mean(s) if reduction == "mean" else T.sum(s)


def cross_entropy(pred, true, reduction="mean"):   # :364-381
    shifted = pred - pred.max().item()             # host sync on the GLOBAL max
    lse = T.log(T.sum(T.exp(shifted), 1, keepdims=True))
    nls = lse - shifted
    if true.ndim == 1:
        nll = nls[range(len(nls.value)), true]
    else:
        nll = nls * true                           # one-hot targets: mean over N*C
    return T.mean(nll) if reduction == "mean" else T.sum(nll)


# ---- norms (nn/modules/norm.py) ---------------------------------------------------------------
def rmsnorm(x, w, eps=1e-6):                       # :221-248
    axes = tuple(-(i + 1) for i in range(w.ndim))
    z = T.square(x).mean(axes, keepdims=True)
    return x / T.sqrt(z + eps) * w


class LayerNormRef:
    """:157-218 -- statistics over the LEADING axes (batch-norm-like), running stats kept."""

    def __init__(self, normalized_shape, eps=1e-6, momentum=0.1, dtype=np.float32):
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.nshape, self.eps, self.momentum = tuple(normalized_shape), eps, momentum
        self.running_mean = np.zeros(self.nshape, dtype)
        self.running_var = np.ones(self.nshape, dtype)
        self.scale = param(np.ones(self.nshape, dtype))
        self.shift = param(np.zeros(self.nshape, dtype))

    def __call__(self, x, train=True):
        if train:
            axis = tuple(range(x.ndim - len(self.nshape)))
            mean = x.mean(axis)
            c = x - mean
            var = T.square(c).mean(axis)
            std = c / T.sqrt(var + self.eps)
            self.running_mean *= (1 - self.momentum); self.running_mean += self.momentum * mean.value
            self.running_var *= (1 - self.momentum); self.running_var += self.momentum * var.value
            return std * self.scale + self.shift
        return (x - self.running_mean) * self.scale / T.sqrt(Var(self.running_var) + self.eps) + self.shift


# ---- recurrent cells (nn/modules/rnn.py) ------------------------------------------------------
def rnn_cell(p, x, h, nonlinearity="tanh"):        # :35-47
    lin = x @ p["Wx"] + h @ p["Wh"]
    if "bias" in p:
        lin = lin + p["bias"]
    return tanh(lin) if nonlinearity == "tanh" else relu(lin)


def gru_cell(p, x, h):                             # :529-544 (not PyTorch's gate algebra)
    lin1 = x @ p["Wx1"] + h @ p["Wh1"]
    if "bias1" in p:
        lin1 = lin1 + p["bias1"]
    z, r = T.split(sigmoid(lin1), 2, axis=1)
    lin2 = x @ p["Wx2"] + (r * h) @ p["Wh2"]
    if "bias2" in p:
        lin2 = lin2 + p["bias2"]
    return (1 - z) * h + z * tanh(lin2)


def gru_sequence(p, x, h0):                        # :623-640, 702-708 (1 layer, unidirectional)
    h, outs = h0, []
    for t in range(x.shape[0]):
        h = gru_cell(p, x[t], h)
        outs.append(T.unsqueeze(h, 0))
    return T.concat(outs), h


# ---- Adam (optim/optimizer.py:160-196) ------------------------------------------------------------
class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = list(params)
        self.lr, (self.b1, self.b2), self.eps, self.wd = lr, betas, eps, weight_decay
        self.m = [np.zeros(p.shape, dtype=p.dtype) for p in self.params]
        self.v = [np.zeros(p.shape, dtype=p.dtype) for p in self.params]
        self.t = 1                                  # starts at 1 (:183)

    def zero_grad(self):
        for p in self.params:
            p.zero_grad()

    def step(self):
        for i, p in enumerate(self.params):
            g = p.grad + self.wd * p.value
            self.m[i] *= self.b1; self.m[i] += (1 - self.b1) * g
            self.v[i] *= self.b2; self.v[i] += (1 - self.b2) * g ** 2
            a_t = math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
            # eps is added to sqrt(v) WITHOUT the bias-correction divisor (:194-195)
            p.value -= self.lr * a_t * self.m[i] / (self.v[i] ** 0.5 + self.eps)
        self.t += 1
