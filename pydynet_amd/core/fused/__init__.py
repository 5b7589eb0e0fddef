"""Fused differentiable nodes of the training hot path.

Each class is an ordinary tape node (same protocol as the reference's operators) that stands
for a chain of generic nodes in the reference and runs as ONE forward and ONE backward HIP
kernel (or GEMM) on a GPU device; on "cpu" the same node evaluates the equivalent NumPy
expression.  Reference chains replaced:

    linear          nn/functional.py:7-11           (matmul + broadcast add; dW summed by the engine)
    linear_relu     nn/functional.py:7-11, 31-32    (Linear -> ReLU of examples/pydynet/mnist.py:70-78: relu and one gradient bit
                                                     per element in the product's store; the consumer's dX product applies the bits)
    rms_norm        nn/modules/norm.py:245-248      (6 nodes)
    silu / swiglu   nn/functional.py:39-40, llm/llama/model.py:56-58
    relu            nn/functional.py:31-32          (maximum(0., x); gradient 1 at x == 0)
    softmax         nn/functional.py:43-49          (last axis)
    rope            llm/llama/model.py:23-44        (26 nodes)
    attention       llm/llama/model.py:112-121      (transpose, matmul, /sqrt(hd), +mask, softmax, matmul); the same chain
                                                    BUILT from plain operators is recognised link by link (chain.py)
    embedding       nn/functional.py:14-20 + tensor.py:937-940 (scatter-ASSIGN gradient)
    cross_entropy   nn/functional.py:364-381        (7 nodes, integer targets)
    linear_cross_entropy  llm/llama/model.py:179 + :239-249 (lm_head -> reshape -> cross entropy as one node); the same
                                                    three calls written with plain operators are taken over as they are
                                                    built (chain.py: loss_chain)
"""
from ._common import (_hip, _L, _contig, hip_f32, _require_f32, _foldable, two_stream, _beside, _is_leaf_f32, _Deferred, _pack_columns, _dx_of_shared_input, _gemm_raw)
from .dense import linear, linear_relu, embedding, cross_entropy, linear_cross_entropy
from .pointwise import gated_sigmoid, swiglu, silu, softmax, rope
from .norm import rms_norm, layer_norm, col_norm
from .attn import _attn_layout, _attn_mask_args, _attn_kernel, attention, qkv_attention
from .ffn import gate_up_swiglu, ffn_swiglu
from .conv import relu, conv2d, conv2d_relu_pool, pool2d
from .recurrent import _cell_grads, rnn_cell, lstm_cell, gru_cell, gru_sequence
from . import chain as _chain_mod
from .chain import attn_link
from .. import tensor as _tensor
_tensor._chain = _chain_mod           # Tensor.__matmul__ / __truediv__ / __add__ consult it (the attention chain of plain operators)
