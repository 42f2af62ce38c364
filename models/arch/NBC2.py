"""NBC2 — drop-in for the reference's models/arch/NBC2.py: narrow-band conformer (NBC2.py:152-289), same constructor arguments,
same forward [B,F,T,dim_input] -> [B,F,T,dim_output], same state_dict keys (`encoder`, `sa_layers.N.{norm1,self_attn,norm2,linear1,
conv.{1,3,4,6},linear2}`, `decoder`).  Each T-F sequence goes through `n_layers` blocks of pre-norm self-attention over time and a
convolutional feed-forward (1x1 -> 3 grouped k=3 convs along T with a norm in the middle -> 1x1).  GroupBatchNorm (:57-145)
normalises with statistics shared by the `group_size` (= num_freqs) sequences of one utterance, in training AND evaluation.
torch.nn modules for training and on the CPU; inference on a HIP device runs the native forward of
nbss_amd/nbc2.py (MFMA tap-GEMMs, attention, GroupBatchNorm and LayerNorm kernels behind the `nbss_nb_*` entry points of the C ABI)."""
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import MultiheadAttention


import os
import warnings
import weakref

_NATIVE = weakref.WeakKeyDictionary()  # NBC2 module -> (nbss_amd.nbc2.NativeNBC2 or None, reason it is None)
_NOTED = weakref.WeakKeyDictionary()   # NBC2 module -> reasons already reported


class LayerNorm(nn.LayerNorm):
    """LayerNorm over the feature axis of [B,T,H] (transpose=False) or [B,H,T] (transpose=True)"""

    def __init__(self, transpose: bool, **kwargs) -> None:
        super().__init__(**kwargs)
        self.transpose = transpose

    def forward(self, input: Tensor) -> Tensor:
        if not self.transpose:
            return super().forward(input)
        return super().forward(input.transpose(-1, -2)).transpose(-1, -2)


class BatchNorm1d(nn.Module):
    def __init__(self, transpose: bool, **kwargs) -> None:
        super().__init__()
        self.transpose = transpose
        self.bn = nn.BatchNorm1d(**kwargs)

    def forward(self, input: Tensor) -> Tensor:  # nn.BatchNorm1d wants [B,H,T]
        return self.bn(input) if self.transpose else self.bn(input.transpose(-1, -2)).transpose(-1, -2)


class GroupNorm(nn.GroupNorm):
    def __init__(self, transpose: bool, **kwargs) -> None:
        super().__init__(**kwargs)
        self.transpose = transpose

    def forward(self, input: Tensor) -> Tensor:
        from models.arch.base.norm import group_norm  # (exact gradients beyond 128 samples on the HIP device: see there)
        gn = lambda v: group_norm(v, self.num_groups, self.weight, self.bias, self.eps)  # noqa: E731
        return gn(input) if self.transpose else gn(input.transpose(-1, -2)).transpose(-1, -2)


class GroupBatchNorm(nn.Module):
    """statistics over (group of `group_size` consecutive batch items) x (feature axis) [x (time axis) when
    share_along_sequence_dim], always computed from the input itself; per-feature affine"""

    def __init__(self, dim_hidden: int, group_size: int, share_along_sequence_dim: bool = False, transpose: bool = False, affine: bool = True,
                 eps: float = 1e-5) -> None:
        super().__init__()
        self.dim_hidden, self.group_size, self.eps, self.affine = dim_hidden, group_size, eps, affine
        self.transpose, self.share_along_sequence_dim = transpose, share_along_sequence_dim
        if affine:
            shape = [dim_hidden, 1] if transpose else [dim_hidden]
            self.weight = nn.Parameter(torch.ones(shape))
            self.bias = nn.Parameter(torch.zeros(shape))

    def forward(self, input: Tensor) -> Tensor:
        Bt = input.shape[0]
        if Bt % self.group_size:
            raise ValueError(f"batch size {Bt} is not divisible by group size {self.group_size}")
        g = input.reshape(Bt // self.group_size, self.group_size, *input.shape[1:])  # [G, gs, T, H] or [G, gs, H, T]
        feat = 2 if self.transpose else 3
        dims = (1, 2, 3) if self.share_along_sequence_dim else (1, feat)
        var, mean = torch.var_mean(g, dim=dims, unbiased=False, keepdim=True)
        out = (g - mean) * torch.rsqrt(var + self.eps)
        if self.affine:
            out = out * self.weight + self.bias
        return out.reshape(input.shape)

    def extra_repr(self) -> str:
        return (f"{self.dim_hidden}, {self.group_size}, share_along_sequence_dim={self.share_along_sequence_dim}, transpose={self.transpose}, "
                f"eps={self.eps}, affine={self.affine}")


def _make_norm(kind: str, dim: int, transpose: bool, n_groups: int, **gbn_kwargs) -> nn.Module:
    if kind == "LN":
        return LayerNorm(normalized_shape=dim, transpose=transpose)
    if kind == "GBN":
        return GroupBatchNorm(dim_hidden=dim, transpose=transpose, **gbn_kwargs)
    if kind == "BN":
        return BatchNorm1d(num_features=dim, transpose=transpose)
    if kind == "GN":
        return GroupNorm(num_groups=n_groups, num_channels=dim, transpose=transpose)
    raise ValueError(f"unknown norm {kind}")


class NBC2Block(nn.Module):
    def __init__(self, dim_hidden: int, dim_ffn: int, n_heads: int, dropout: float = 0, conv_kernel_size: int = 3, n_conv_groups: int = 8,
                 norms: Tuple[str, str, str] = ("LN", "GBN", "GBN"),
                 group_batch_norm_kwargs: Dict[str, Any] = {"group_size": 257, "share_along_sequence_dim": False}) -> None:
        super().__init__()
        gk = dict(group_batch_norm_kwargs)
        self.norm1 = _make_norm(norms[0], dim_hidden, False, n_conv_groups, **gk)
        self.self_attn = MultiheadAttention(embed_dim=dim_hidden, num_heads=n_heads, batch_first=True)
        self.dropout1 = nn.Dropout(dropout)
        self.norm2 = _make_norm(norms[1], dim_hidden, False, n_conv_groups, **gk)
        self.linear1 = nn.Linear(dim_hidden, dim_ffn)

        def gconv():
            return nn.Conv1d(dim_ffn, dim_ffn, kernel_size=conv_kernel_size, padding="same", groups=n_conv_groups, bias=True)

        self.conv = nn.Sequential(nn.SiLU(), gconv(), nn.SiLU(), gconv(), _make_norm(norms[2], dim_ffn, True, n_conv_groups, **gk), nn.SiLU(), gconv(),
                                  nn.SiLU(), nn.Dropout(dropout))
        self.linear2 = nn.Linear(dim_ffn, dim_hidden)
        self.dropout2 = nn.Dropout(dropout)
        for lin in (self.linear1, self.linear2):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, x: Tensor, att_mask: Optional[Tensor] = None):
        """x [batch, seq, feature] -> (x, attention [batch, head, seq, seq])"""
        u = self.norm1(x)
        a, attn = self.self_attn(u, u, u, average_attn_weights=False, attn_mask=att_mask)
        x = x + self.dropout1(a)
        h = self.linear1(self.norm2(x)).transpose(-1, -2)  # [B, ffn, T] for the convs along time
        x = x + self.dropout2(self.linear2(self.conv(h).transpose(-1, -2)))
        return x, attn


class NBC2(nn.Module):
    def __init__(self, dim_input: int, dim_output: int, n_layers: int, encoder_kernel_size: int = 5, dim_hidden: int = 192, dim_ffn: int = 384,
                 num_freqs: int = 257,
                 block_kwargs: Dict[str, Any] = {"n_heads": 2, "dropout": 0, "conv_kernel_size": 3, "n_conv_groups": 8, "norms": ("LN", "GBN", "GBN"),
                                                 "group_batch_norm_kwargs": {"share_along_sequence_dim": False}}):
        super().__init__()
        bk = dict(block_kwargs)
        bk["group_batch_norm_kwargs"] = dict(bk.get("group_batch_norm_kwargs", {}), group_size=num_freqs)  # one group = one utterance's bins
        self.encoder = nn.Conv1d(dim_input, dim_hidden, kernel_size=encoder_kernel_size, stride=1, padding="same")
        self.sa_layers = nn.ModuleList([NBC2Block(dim_hidden=dim_hidden, dim_ffn=dim_ffn, **bk) for _ in range(n_layers)])
        self.decoder = nn.Linear(dim_hidden, dim_output)

    def _native(self):
        """the HIP path (nbss_amd/nbc2.py) when this configuration is one its kernels are built for, else None.  The handle lives in a module-level
        WeakKeyDictionary (a ctypes library handle as a module attribute would break deepcopy / pickle of the module); a library that cannot be
        loaded means the torch.nn path, not an exception — the reason is kept and reported once by forward()."""
        if self not in _NATIVE:
            runner, why = None, None
            try:
                from nbss_amd._lib import hip
                from nbss_amd.nbc2 import NativeNBC2, supported
                why = supported(self)
                if why is None:
                    runner = NativeNBC2(self, hip())
            except Exception as e:  # (no library / no HIP runtime: torch.nn below)
                runner, why = None, f"{type(e).__name__}: {e}"
            _NATIVE[self] = (runner, why)
        return _NATIVE[self][0]

    def _torch_path_note(self, why: str) -> None:
        """one warning per module and reason: a user on a HIP device can tell which path ran"""
        seen = _NOTED.setdefault(self, set())
        if why not in seen:
            seen.add(why)
            warnings.warn(f"NBC2: torch.nn path instead of the native HIP kernels ({why})", RuntimeWarning, stacklevel=3)

    def _native_or_reason(self, x: Tensor):
        """(runner, None) when the native path takes this call, else (None, reason); reason None on the CPU (nothing to report)"""
        if not x.is_cuda:
            return None, None
        B, F, T, _ = x.shape
        if os.environ.get("NBSS_NBC2_NATIVE", "1") == "0":  # (the same switch NBC and NB-BLSTM have: A/B runs, FLOP counting on the torch.nn modules)
            return None, "NBSS_NBC2_NATIVE=0"
        if x.dtype not in (torch.float32, torch.bfloat16):
            return None, f"input dtype {x.dtype}"
        nat = self._native()  # first: supported() also guards the attribute reads below (other norm types have no group_size)
        if nat is None:
            return None, _NATIVE[self][1] or "native path unavailable"
        if T > 256:
            return None, f"{T} frames: the attention kernels keep a sequence in LDS (<= 256)"
        gs = getattr(self.sa_layers[0].norm2, "group_size", None)
        if F != gs:
            return None, f"{F} frequencies per utterance, GroupBatchNorm group_size {gs}"
        return nat, None

    def forward(self, x: Tensor) -> Tensor:
        B, F, T, _ = x.shape
        # on a HIP device: the native forward (torch.no_grad(): validate / test / predict) or the native training path (autograd.Function whose backward
        # runs the nbss_nb_*_bwd building blocks); CPU and configurations the kernels are not built for: the torch.nn modules below (with one warning
        # naming the reason when the input is on a device)
        nat, why = self._native_or_reason(x)
        if nat is not None:
            if not torch.is_grad_enabled():
                return nat.forward(x.contiguous())
            from nbss_amd.nbc2 import _train_supported
            why = _train_supported(self)
            if why is None and x.requires_grad:
                why = "the input requires a gradient (the native backward produces parameter gradients only)"
            if why is None:
                return nat.forward_train(x.contiguous())
        if why is not None:
            self._torch_path_note(why)
        h = self.encoder(x.reshape(B * F, T, -1).transpose(1, 2)).transpose(1, 2)
        for block in self.sa_layers:
            h, _ = block(h)
        return self.decoder(h).reshape(B, F, T, -1).contiguous()
