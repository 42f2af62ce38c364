"""Normalisation layers of the SpatialNet family — drop-in for the reference's models/arch/base/norm.py
(LayerNorm :11-27, GroupNorm :79-91, factory new_norm :232-247).

Inside models.arch.SpatialNet these objects are PARAMETER HOLDERS (their tensors are read by the HIP
kernels, which fuse the normalisation into the surrounding block); their own forward() is plain PyTorch
and only serves stand-alone use of the class.  Only the types configs/SpatialNet.yaml can reach ("LN",
"GN") are provided: the reference's other branches are broken upstream and no shipped
config selects them.
"""
from torch import Tensor, nn


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
        o = super().forward(input)
        return o if self.seq_last else o.transpose(-1, 1)


def new_norm(norm_type: str, dim_hidden: int, seq_last: bool, group_size: int = None, num_groups: int = None) -> nn.Module:
    if norm_type.upper() == "LN":
        return LayerNorm(seq_last=seq_last, normalized_shape=dim_hidden)
    if norm_type.upper() == "GN":
        return GroupNorm(seq_last=seq_last, num_groups=num_groups, num_channels=dim_hidden)
    raise NotImplementedError(f"norm type {norm_type!r}: only 'LN' and 'GN' have MI355X kernels (the types configs/SpatialNet.yaml uses)")
