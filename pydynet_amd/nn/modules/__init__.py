from .activation import Sigmoid, Tanh, ReLU, LeakyReLU, Softmax
from .norm import BatchNorm1d, BatchNorm2d, LayerNorm, RMSNorm
from .conv import Conv1d, Conv2d
from .pool import MaxPool1d, MaxPool2d, AvgPool1d, AvgPool2d
from .dropout import Dropout
from .linear import Linear, Embedding
from .loss import MSELoss, NLLLoss, CrossEntropyLoss
from .module import Module, Sequential, ModuleList
from .rnn import RNN, LSTM, GRU, RNNCell, LSTMCell, GRUCell

__all__ = ["Sigmoid", "Tanh", "ReLU", "LeakyReLU", "Softmax", "BatchNorm1d", "BatchNorm2d", "LayerNorm",
           "RMSNorm", "Conv1d", "Conv2d", "MaxPool1d", "MaxPool2d", "AvgPool1d", "AvgPool2d", "Dropout",
           "Linear", "Embedding", "MSELoss", "NLLLoss", "CrossEntropyLoss", "Module", "Sequential",
           "ModuleList", "RNN", "LSTM", "GRU", "RNNCell", "LSTMCell", "GRUCell"]
