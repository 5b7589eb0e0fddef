"""Functional layer API (surface of pydynet/nn/functional.py).  Where the reference composes
several generic nodes, the same mathematical function is issued as ONE fused node
(pydynet_amd/core/fused/) -- on a HIP device that is one hand-written kernel per direction."""
import numpy as np

from ..core import tensor, function, fused
from ..core.tensor import Tensor
from ..core.function import unsqueeze
from ..autograd import no_grad


def linear(x, weight, bias):
    if (x.ndim >= 2 and weight.ndim == 2 and x.dtype == weight.dtype and np.issubdtype(x.dtype, np.floating)
            and fused.hip_f32(x, weight, bias)):
        return fused.linear(x, weight, bias)
    affine = x @ weight
    return affine + bias if bias is not None else affine


def embedding(x, weight, padding_idx):
    if weight.ndim == 2 and weight.dtype == np.float32:
        query = fused.embedding(x, weight)
    else:
        query = weight[x]
    if padding_idx is not None:
        with no_grad():
            mask = unsqueeze(x.ne(padding_idx), -1)
        query = query * mask
    return query


def sigmoid(x): return tensor.sigmoid(x)
def tanh(x): return tensor.tanh(x)
def relu(x):
    if type(x) is fused.linear and x._pending is not None:
        # relu(linear(...)) with the product still deferred: ONE node, relu and its gradient bits in the GEMM stores
        ins = x._pending
        return fused.linear_relu(ins[0], ins[1], ins[2] if x.has_bias else None)
    return fused.relu(x)                    # (generic kernels inside for non-float32 HIP operands)
def leaky_relu(x, alpha: float): return tensor.maximum(x, alpha * x)


def silu(x):
    if x.dtype == np.float32:
        return fused.silu(x)
    return x / (1 + tensor.exp(-x))


def softmax(x, axis=None):
    if axis is not None and x.ndim >= 1 and axis in (-1, x.ndim - 1) and x.dtype == np.float32:
        if x._pending_link:                  # a deferred link of the plain-operator attention chain (core/fused/chain.py)
            r = fused.chain.on_softmax(x)
            if r is not None:
                return r
        return fused.softmax(x)
    with no_grad():
        max_ = x.max(axis, keepdims=True)
    exp_ = tensor.exp(x - max_)
    return exp_ / tensor.sum(exp_, axis=axis, keepdims=True)


def log_softmax(x, axis=None, keepdims=False):
    with no_grad():
        max_ = x.max(axis, keepdims=True)
    shifted = x - max_
    return shifted - tensor.log(tensor.sum(tensor.exp(shifted), axis=axis, keepdims=keepdims))


def conv2d(x, kernel, padding: int = 0, stride: int = 1, bias=None):
    """x (N, C, H, W), kernel (O, C, k, k); square kernel / stride / padding only."""
    if not fused.hip_f32(x, kernel, bias):
        raise TypeError("conv2d on a HIP device is float32-only (im2col / MFMA GEMM kernels): got "
                        f"{x.dtype} / {kernel.dtype}; pass dtype=np.float32 or stay on the cpu device")
    return fused.conv2d(x, kernel, bias, padding, stride)


def _pool2d(x, kernel_size, stride, padding, mode):
    if (mode == "max" and kernel_size == 2 and stride == 2 and padding == 0 and type(x) is fused.relu
            and x._pending is not None and x._pending[0]._pending is not None):
        # max_pool2d(relu(conv2d(...)), 2, 2) with both producers still deferred: ONE kernel (fused._Deferred)
        conv = x._pending[0]
        ins = conv._pending
        return fused.conv2d_relu_pool(ins[0], ins[1], ins[2] if conv.has_bias else None, conv.padding, conv.stride)
    if not fused.hip_f32(x):
        raise TypeError(f"{mode}_pool2d on a HIP device is float32-only: got {x.dtype}")
    return fused.pool2d(x, kernel_size, stride, padding, mode)


def max_pool2d(x, kernel_size: int, stride: int, padding=0):
    return _pool2d(x, kernel_size, stride, padding, "max")


def avg_pool2d(x, kernel_size: int, stride: int, padding=0):
    return _pool2d(x, kernel_size, stride, padding, "avg")


def _as4d(x):
    return x.reshape(x.shape[0], x.shape[1], 1, x.shape[2])


def conv1d(x, kernel, padding: int = 0, stride: int = 1):
    """1-D convolution on the generic operators (window gather by slicing + matmul).  Documented
    deviation: the reference's conv1d contracts the output-position axis of its im2col buffer with the
    kernel axis (`col @ kernel.transpose(1, 2, 0)`, nn/functional.py:139) and therefore raises for every
    shape with n_output != kernel_size; this is the convolution that expression was meant to be."""
    N, C, F = x.shape
    O, _, k = kernel.shape
    if padding:
        zeros = Tensor(np.zeros((N, C, padding)), dtype=x.dtype, device=x.device)
        x = tensor.concat([zeros, x, zeros], axis=2)
        F = F + 2 * padding
    n_out = (F - k) // stride + 1
    cols = [function.unsqueeze(x[:, :, i:i + stride * (n_out - 1) + 1:stride], 2) for i in range(k)]
    col = tensor.concat(cols, axis=2)                       # (N, C, k, n_out)
    return (col.transpose(0, 1, 3, 2) @ kernel.transpose(1, 2, 0)).sum(1).swapaxes(1, 2)


def _pool1d(x, kernel_size, stride, padding, reducer):
    """Windows of the zero-padded input as (N, C, kernel_size, n_out) -- the reference's im2col1d layout
    (nn/functional.py:61-84) -- reduced over the LAST axis exactly as the reference does
    (`col.max(-1)` / `col.mean(-1)`, :165, :191).  Reference quirk kept for parity: that axis is the
    output position, not the window, so the result is (N, C, kernel_size): entry j is the max / mean of
    the j-th element of every window.  (The 2-D pooling reduces over the window as usual.)"""
    N, C, F = x.shape
    if padding:
        zeros = Tensor(np.zeros((N, C, padding)), dtype=x.dtype, device=x.device)
        x = tensor.concat([zeros, x, zeros], axis=2)
        F = F + 2 * padding
    n_out = (F - kernel_size) // stride + 1
    cols = [function.unsqueeze(x[:, :, i:i + stride * (n_out - 1) + 1:stride], 2) for i in range(kernel_size)]
    return reducer(tensor.concat(cols, axis=2))               # (N, C, k, n_out) -> (N, C, k)


def max_pool1d(x, kernel_size, stride, padding=0):
    return _pool1d(x, kernel_size, stride, padding, lambda c: c.max(-1))


def avg_pool1d(x, kernel_size, stride, padding=0):
    return _pool1d(x, kernel_size, stride, padding, lambda c: c.mean(-1))


def _reduce(value, reduction):
    if reduction == 'mean':
        return tensor.mean(value)
    if reduction == 'sum':
        return tensor.sum(value)
    raise ValueError("reduction must be mean or sum.")


def mse_loss(y_pred, y_true, reduction='mean'):
    return _reduce(function.square(y_pred - y_true), reduction)


def nll_loss(y_pred, y_true, reduction='mean'):
    return _reduce(-y_pred * y_true, reduction)


def cross_entropy_loss(y_pred, y_true, reduction='mean'):
    if reduction not in ('mean', 'sum'):
        raise ValueError("reduction must be mean or sum.")
    if y_true.ndim == 1 and y_pred.ndim == 2 and y_pred.dtype == np.float32:
        # a projection that has not run yet + this loss = one node (model.py:239-249 written with plain operators)
        r = fused.chain.on_cross_entropy(y_pred, y_true, reduction)
        return r if r is not None else fused.cross_entropy(y_pred, y_true, reduction)
    # one-hot / soft targets: the reference's generic chain, including its mean over N*C
    shifted = y_pred - y_pred.max().item()
    log_sum_exp = tensor.log(tensor.sum(tensor.exp(shifted), 1, keepdims=True))
    neg_log_sm = log_sum_exp - shifted
    nll = neg_log_sm[range(len(neg_log_sm)), y_true] if y_true.ndim == 1 else neg_log_sm * y_true
    return _reduce(nll, reduction)
