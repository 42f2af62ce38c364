"""Normalisation layers of the SpatialNet family — drop-in for the reference's models/arch/base/norm.py
(LayerNorm :11-27, GroupNorm :79-91, factory new_norm :232-247).

Inside models.arch.SpatialNet these objects are PARAMETER HOLDERS (their tensors are read by the HIP
kernels, which fuse the normalisation into the surrounding block); their own forward() is plain PyTorch
and only serves stand-alone use of the class.  Only the types configs/SpatialNet.yaml can reach ("LN",
"GN") are provided: the reference's other branches are broken upstream and no shipped
config selects them.
"""
import torch
from torch import Tensor, nn


def group_norm(input: Tensor, num_groups: int, weight: Tensor, bias: Tensor, eps: float) -> Tensor:
    """torch.nn.functional.group_norm, except on a HIP device with more than 128 samples, where torch 2.10+rocm7.0's native backward
    returns wrong weight / bias gradients in every dtype (tests/diag/torch_group_norm_check.py: rel. error ~1 from 129 samples on, exact
    up to 128) — every training batch of the narrow-band / online models is far beyond that (one sample per frequency or frame).
    The written-out form below runs on plain elementwise / reduction kernels and is exact everywhere."""
    if not (input.is_cuda and input.shape[0] > 128 and torch.is_grad_enabled()):
        return nn.functional.group_norm(input, num_groups, weight, bias, eps)
    N, C = input.shape[0], input.shape[1]
    hg = input.reshape(N, num_groups, -1).float()
    mean = hg.mean(-1, keepdim=True)
    var = hg.var(-1, unbiased=False, keepdim=True)
    xhat = ((hg - mean) * torch.rsqrt(var + eps)).reshape(input.shape)
    shape = (1, C) + (1,) * (input.dim() - 2)
    out = xhat * weight.float().reshape(shape) + bias.float().reshape(shape) if weight is not None else xhat
    return out.to(input.dtype)


class LayerNorm(nn.LayerNorm):
    def __init__(self, seq_last: bool, **kwargs) -> None:
        super().__init__(**kwargs)
        self.seq_last = seq_last

    def forward(self, input: Tensor) -> Tensor:
        if self.seq_last:
            input = input.transpose(-1, 1)
        o = super().forward(input)
        return o.transpose(-1, 1) if self.seq_last else o


class GroupNorm(nn.GroupNorm):
    def __init__(self, seq_last: bool, **kwargs) -> None:
        super().__init__(**kwargs)
        self.seq_last = seq_last

    def forward(self, input: Tensor) -> Tensor:
        if not self.seq_last:
            input = input.transpose(-1, 1)
        o = group_norm(input, self.num_groups, self.weight, self.bias, self.eps)
        return o if self.seq_last else o.transpose(-1, 1)


def new_norm(norm_type: str, dim_hidden: int, seq_last: bool, group_size: int = None, num_groups: int = None) -> nn.Module:
    if norm_type.upper() == "LN":
        return LayerNorm(seq_last=seq_last, normalized_shape=dim_hidden)
    if norm_type.upper() == "GN":
        return GroupNorm(seq_last=seq_last, num_groups=num_groups, num_channels=dim_hidden)
    raise NotImplementedError(f"norm type {norm_type!r}: only 'LN' and 'GN' have MI355X kernels (the types configs/SpatialNet.yaml uses)")
