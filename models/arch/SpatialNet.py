"""SpatialNet — drop-in for the reference's models/arch/SpatialNet.py (same constructor arguments
:154-171, same forward(x, return_attn_score=False) :202-220, same state_dict keys), executed by the
MI355X-native HIP kernels of nbss_amd (no PyTorch ops on the hot path, no CPU fallback).

The torch.nn sub-modules below exist to hold the parameters under the reference's names and with
torch.nn's default initialisation; forward() never calls them.  All parameters are re-pointed into ONE
flat fp32 buffer (nbss_amd.engine.SpatialNetEngine) that the kernels, the fused optimizer and the
gradient all-reduce work on.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import MultiheadAttention

from models.arch.base.linear_group import LinearGroup
from models.arch.base.norm import new_norm


class SpatialNetLayer(nn.Module):
    """parameter container of one layer (reference :12-73); executed block-wise by nbss_amd"""

    def __init__(self, dim_hidden: int, dim_ffn: int, dim_squeeze: int, num_freqs: int, num_heads: int, dropout: Tuple[float, float, float] = (0, 0, 0),
                 kernel_size: Tuple[int, int] = (5, 3), conv_groups: Tuple[int, int] = (8, 8), norms: List[str] = ("LN", "LN", "GN", "LN", "LN", "LN"),
                 padding: str = "zeros", full: nn.Module = None) -> None:
        super().__init__()
        fg, tg = conv_groups
        fk, tk = kernel_size
        self.fconv1 = nn.ModuleList([
            new_norm(norms[3], dim_hidden, seq_last=True, group_size=None, num_groups=fg),
            nn.Conv1d(dim_hidden, dim_hidden, kernel_size=fk, groups=fg, padding="same", padding_mode=padding),
            nn.PReLU(dim_hidden),
        ])
        self.norm_full = new_norm(norms[5], dim_hidden, seq_last=False, group_size=None, num_groups=fg)
        self.full_share = full is not None
        self.squeeze = nn.Sequential(nn.Conv1d(dim_hidden, dim_squeeze, kernel_size=1), nn.SiLU())
        self.full = LinearGroup(num_freqs, num_freqs, num_groups=dim_squeeze) if full is None else full
        self.unsqueeze = nn.Sequential(nn.Conv1d(dim_squeeze, dim_hidden, kernel_size=1), nn.SiLU())
        self.fconv2 = nn.ModuleList([
            new_norm(norms[4], dim_hidden, seq_last=True, group_size=None, num_groups=fg),
            nn.Conv1d(dim_hidden, dim_hidden, kernel_size=fk, groups=fg, padding="same", padding_mode=padding),
            nn.PReLU(dim_hidden),
        ])
        self.norm_mhsa = new_norm(norms[0], dim_hidden, seq_last=False, group_size=None, num_groups=tg)
        self.mhsa = MultiheadAttention(embed_dim=dim_hidden, num_heads=num_heads, batch_first=True)
        self.tconvffn = nn.ModuleList([
            new_norm(norms[1], dim_hidden, seq_last=True, group_size=None, num_groups=tg),
            nn.Conv1d(dim_hidden, dim_ffn, kernel_size=1),
            nn.SiLU(),
            nn.Conv1d(dim_ffn, dim_ffn, kernel_size=tk, padding="same", groups=tg),
            nn.SiLU(),
            nn.Conv1d(dim_ffn, dim_ffn, kernel_size=tk, padding="same", groups=tg),
            new_norm(norms[2], dim_ffn, seq_last=True, group_size=None, num_groups=tg),
            nn.SiLU(),
            nn.Conv1d(dim_ffn, dim_ffn, kernel_size=tk, padding="same", groups=tg),
            nn.SiLU(),
            nn.Conv1d(dim_ffn, dim_hidden, kernel_size=1),
        ])

    def extra_repr(self) -> str:
        return f"full_share={self.full_share}"


class _SpatialNetFn(torch.autograd.Function):
    """forward / backward of the whole network as ONE autograd node (two C calls)"""

    @staticmethod
    def forward(ctx, module, x, *params):
        eng, dtype = module._engine, module._stream_dtype()
        xin = x.detach().to(eng.stream_dtype(dtype)).contiguous()
        train = module._want_grad  # grad mode is always off inside Function.forward: the module samples it before apply()
        out = eng.forward(xin, train=train, dtype=dtype)
        if train:
            eng.forward_serial = getattr(eng, "forward_serial", 0) + 1
        ctx.module, ctx.dtype, ctx.train, ctx.serial = module, dtype, train, getattr(eng, "forward_serial", 0)
        ctx.save_for_backward(xin)
        return out

    @staticmethod
    def backward(ctx, dout):
        module = ctx.module
        if not ctx.train:
            raise RuntimeError("SpatialNet: backward through a forward that ran without gradient tracking")
        (xin,) = ctx.saved_tensors
        eng = module._engine
        if ctx.serial != getattr(eng, "forward_serial", 0):
            # the engine keeps the activations of its LAST train-mode forward only
            raise RuntimeError("SpatialNet: backward of a forward whose saved activations were overwritten by a later forward "
                               "(run forward -> backward pairs one at a time)")
        eng.grads.zero_()
        eng.backward(xin, dout.contiguous().float(), dtype=ctx.dtype)
        g = eng.grads.clone()
        grads = [g[off:off + p.numel()].view_as(p) for (p, off) in zip(module._flat_params, module._flat_offsets)]
        return (None, None, *grads)  # the network input needs no gradient (SharedTrainer feeds STFT coefficients)


class SpatialNet(nn.Module):
    def __init__(self, dim_input: int, dim_output: int, dim_squeeze: int, num_layers: int, num_freqs: int, encoder_kernel_size: int = 5,
                 dim_hidden: int = 192, dim_ffn: int = 384, num_heads: int = 2, dropout: Tuple[float, float, float] = (0, 0, 0),
                 kernel_size: Tuple[int, int] = (5, 3), conv_groups: Tuple[int, int] = (8, 8), norms: List[str] = ("LN", "LN", "GN", "LN", "LN", "LN"),
                 padding: str = "zeros", full_share: int = 0):
        super().__init__()
        if any(d > 0 for d in dropout):
            raise NotImplementedError("dropout > 0 has no MI355X kernel (every shipped SpatialNet config uses dropout 0)")
        if padding != "zeros" or tuple(n.upper() for n in norms) != ("LN", "LN", "GN", "LN", "LN", "LN"):
            raise NotImplementedError("only padding='zeros' and norms=(LN,LN,GN,LN,LN,LN) have MI355X kernels")
        self.hp = dict(dim_input=dim_input, dim_output=dim_output, num_freqs=num_freqs, num_layers=num_layers, dim_hidden=dim_hidden, dim_ffn=dim_ffn,
                       dim_squeeze=dim_squeeze, num_heads=num_heads, encoder_kernel_size=encoder_kernel_size, kernel_size=tuple(kernel_size),
                       conv_groups=tuple(conv_groups), full_share=full_share)
        self.encoder = nn.Conv1d(dim_input, dim_hidden, kernel_size=encoder_kernel_size, stride=1, padding="same")
        full = None
        layers = []
        for l in range(num_layers):
            layer = SpatialNetLayer(dim_hidden=dim_hidden, dim_ffn=dim_ffn, dim_squeeze=dim_squeeze, num_freqs=num_freqs, num_heads=num_heads,
                                    dropout=dropout, kernel_size=kernel_size, conv_groups=conv_groups, norms=norms, padding=padding,
                                    full=full if l > full_share else None)
            full = layer.full
            layers.append(layer)
        self.layers = nn.ModuleList(layers)
        self.decoder = nn.Linear(dim_hidden, dim_output)
        self._engine = None
        self._flat_params: List[nn.Parameter] = []
        self._flat_offsets: List[int] = []

    # ---- flat-buffer plumbing ---------------------------------------------------------------------
    def _stream_dtype(self) -> int:
        from nbss_amd._lib import NBSS_BF16, NBSS_F32
        if torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16:
            return NBSS_BF16  # Lightning 'bf16-mixed' == autocast(bfloat16): bf16 stream, fp32 master weights
        return NBSS_F32

    def _engine_for(self, device):
        """(re)bind every parameter into the engine's flat fp32 buffer on `device`"""
        from nbss_amd._lib import hip
        from nbss_amd.engine import SpatialNetEngine
        if device.type != "cuda":
            raise RuntimeError("models.arch.SpatialNet runs on MI355X HIP kernels only: move the module and its input to a 'cuda' (HIP) device")
        named = dict(self.named_parameters(remove_duplicate=False))
        eng = self._engine
        bound = eng is not None and eng.device == device and all(
            named[k].data_ptr() == eng.params.data_ptr() + 4 * off for k, (off, _) in list(eng.table.items())[:3])
        if not bound:
            eng = SpatialNetEngine(hip(), device, **self.hp)
            eng.load_params({k: named[k].detach() for k in eng.table})
            views = eng.param_views(eng.params)
            seen, self._flat_params, self._flat_offsets = set(), [], []
            for k, (off, _) in eng.table.items():
                p = named[k]
                if id(p) in seen:
                    continue
                seen.add(id(p))
                p.data = views[k]
                self._flat_params.append(p)
                self._flat_offsets.append(off)
            self._engine = eng
        eng.version += 1  # parameters may have been updated in place by an external optimizer: re-pack
        return eng

    def forward(self, x: Tensor, return_attn_score: bool = False):
        """x [B, F, T, dim_input] -> [B, F, T, dim_output] (fp32).  Attention maps are never materialised by the
        fused kernel; like the reference (whose `need_weights` is always False, SpatialNet.py:97) the list holds None."""
        if x.dim() != 4 or x.shape[1] != self.hp["num_freqs"] or x.shape[3] != self.hp["dim_input"]:
            raise ValueError(f"expected [B, {self.hp['num_freqs']}, T, {self.hp['dim_input']}], got {tuple(x.shape)}")
        self._engine_for(x.device)
        self._want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._flat_params)
        y = _SpatialNetFn.apply(self, x, *self._flat_params)
        if return_attn_score:
            return y.contiguous(), [None] * len(self.layers)
        return y.contiguous()
