"""LinearGroup — drop-in for models/arch/base/linear_group.py:7-37: one (out x in) matrix per group,
y[..., g, k] = sum_h x[..., g, h] W[g, k, h] + b[g, k].  In SpatialNet it is the full-band linear along F
(one F x F matrix per squeeze channel, shared across layers) and is executed by the fused `full` HIP
kernel; the PyTorch forward below is for stand-alone use only."""
import math

import torch
from torch import Tensor, nn
from torch.nn import init
from torch.nn.parameter import Parameter


class LinearGroup(nn.Module):
    def __init__(self, in_features: int, out_features: int, num_groups: int, bias: bool = True) -> None:
        super().__init__()
        self.in_features, self.out_features, self.num_groups = in_features, out_features, num_groups
        self.weight = Parameter(torch.empty((num_groups, out_features, in_features)))
        if bias:
            self.bias = Parameter(torch.empty(num_groups, out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:  # nn.Linear's default initialisation
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)

    def forward(self, x: Tensor) -> Tensor:
        y = torch.einsum("...gh,gkh->...gk", x, self.weight)
        return y if self.bias is None else y + self.bias

    def extra_repr(self) -> str:
        return f"{self.in_features}, {self.out_features}, num_groups={self.num_groups}, bias={self.bias is not None}"
