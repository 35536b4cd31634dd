"""Stand-alone layer ops -- host-side mirror of neddf/nn_module/ (forward halves).

In the reference these classes ARE the network (NeDDF.forward chains them as
eager torch ops).  Here the network is the fused field kernels; these wrappers
expose the same device code one op at a time (C ABI: neddf_op_*), mainly so
each op has a drop-in counterpart and a unit-level parity test.  Forward only:
the hand-written backward passes are training code.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from ._lib import Context
from .network import PositionalEncodingInfo

_RELU, _LEAKY, _TANHEXP, _SOFTPLUS, _SIGMOID = range(5)


def _op(kind: int, x: Tensor, J: Optional[Tensor] = None):
    return Context.get(x.device).op_activation(kind, x, J)


class _GradFunction:
    kind = _RELU

    @classmethod
    def apply(cls, x: Tensor, J: Tensor) -> Tuple[Tensor, Tensor]:
        """(x [N,C], J [N,3,C]) -> (a(x), a'(x) * J)"""
        return _op(cls.kind, x, J)


class ReLUGradFunction(_GradFunction):          # with_grad/relu.py:15-40
    kind = _RELU


class LeakyReLUGradFunction(_GradFunction):     # with_grad/leaky_relu.py:15-41
    kind = _LEAKY


class TanhExpGradFunction(_GradFunction):       # with_grad/tanh_exp.py:15-54
    kind = _TANHEXP


class SoftplusGradFunction(_GradFunction):      # with_grad/softplus.py:15-52
    kind = _SOFTPLUS


class SigmoidGradFunction(_GradFunction):       # with_grad/sigmoid.py:15-46 (one channel)
    kind = _SIGMOID


class tanhExp:                                  # nn_module/tanh_exp.py:15-33
    @staticmethod
    def apply(x: Tensor) -> Tensor:
        shape = x.shape
        return _op(_TANHEXP, x.reshape(-1, shape[-1])).reshape(shape)


class LinearGradFunction:                       # with_grad/linear.py:15-46
    @staticmethod
    def apply(x: Tensor, J: Tensor, weight_t: Tensor, bias: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        return Context.get(x.device).op_linear_grad(x, J, weight_t, bias)


class LinearGradLayer(nn.Module):               # with_grad/linear.py:87-133
    def __init__(self, input_ch: int = 128, output_ch: int = 128) -> None:
        super().__init__()
        self.input_ch, self.output_ch = input_ch, output_ch
        self.weight = nn.Parameter(torch.empty(input_ch, output_ch))
        self.bias = nn.Parameter(torch.zeros(output_ch))
        nn.init.xavier_normal_(self.weight)

    def forward(self, x: Tensor, J: Tensor) -> Tuple[Tensor, Tensor]:
        return LinearGradFunction.apply(x, J, self.weight, self.bias)


class PositionalEncoding(PositionalEncodingInfo):           # nn_module/positional_encoding.py:8-65
    def forward(self, x: Tensor, scale: Optional[Tensor] = None) -> Tensor:
        return Context.get(x.device).op_positional_encoding(x, None, scale, self.embed_dim)


class PositionalEncodingGradLayer(PositionalEncodingInfo):  # with_grad/positional_encoding.py:8-87
    def forward(self, x: Tensor, J: Tensor, scale: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        return Context.get(x.device).op_positional_encoding(x, J, scale, self.embed_dim)

    def withoutGrad(self, x: Tensor, scale: Optional[Tensor] = None) -> Tensor:
        return Context.get(x.device).op_positional_encoding(x, None, scale, self.embed_dim)
