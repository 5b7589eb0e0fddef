from .tensor import (Tensor, Graph, add, sub, mul, div, pow, matmul, abs, sum, mean, min, max,
                     argmax, argmin, maximum, minimum, exp, log, sign, reshape, transpose, swapaxes,
                     concat, sigmoid, tanh, _UnaryOperator, _BinaryOperator, _Operator)
from .function import sqrt, square, vsplit, hsplit, dsplit, split, unsqueeze, squeeze
