"""Native forward of the narrow-band conformer NBC2 (reference: models/arch/NBC2.py:152-289) on the HIP device, sequenced from the
geometry-generic building blocks of the C ABI (`nbss_nb_*`, csrc/gbwd.hip): the encoder Conv1d along time, per block LayerNorm ->
in_proj -> softmax(q k^T / sqrt(dh)) v per (sequence, head) -> out_proj + residual, GroupBatchNorm -> Linear -> SiLU -> grouped conv ->
SiLU -> grouped conv -> GroupBatchNorm -> SiLU -> grouped conv -> SiLU -> Linear + residual, and the decoder.  Every GEMM-shaped step is
one MFMA tap-GEMM launch (weights re-laid on the fly from the module's own fp32 parameters); activations stay in the
[B*F, T, C] layout of the reference.  Inference only (`torch.no_grad()` / eval): training of NBC2 runs its torch.nn modules.

`supported(net)` names what the kernels are built for: norms (LN, GBN, GBN) with per-frame GroupBatchNorm statistics, no dropout, head
width 24 or 48, channel counts that are multiples of 8 per conv group, sequences of at most 256 frames."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import ops
from ._lib import NBSS_BF16, NBSS_F32, Lib, NbssError


def supported(net) -> Optional[str]:
    """None when `net` (models.arch.NBC2.NBC2) can run through the native forward, else the reason"""
    from models.arch.NBC2 import GroupBatchNorm, LayerNorm
    blocks = list(net.sa_layers)
    if not blocks:
        return "no layers"
    H, FFN = net.encoder.out_channels, blocks[0].linear1.out_features
    if net.encoder.kernel_size[0] % 2 == 0 or net.encoder.groups != 1:
        return "encoder must be an odd-kernel dense Conv1d"
    for b in blocks:
        if not isinstance(b.norm1, LayerNorm) or not isinstance(b.norm2, GroupBatchNorm) or not isinstance(b.conv[4], GroupBatchNorm):
            return "norms must be (LN, GBN, GBN)"
        if b.norm2.share_along_sequence_dim or b.conv[4].share_along_sequence_dim:
            return "GroupBatchNorm statistics must be per frame (share_along_sequence_dim = False)"
        if b.dropout1.p or b.dropout2.p or b.conv[8].p:
            return "dropout must be 0"
        if H // b.self_attn.num_heads not in (24, 48) or not b.self_attn._qkv_same_embed_dim or b.self_attn.in_proj_bias is None:
            return "attention head width must be 24 or 48 (packed in_proj with bias)"
        g = b.conv[1].groups
        if (FFN // g) % 8 or FFN % g or b.conv[1].kernel_size[0] % 2 == 0:
            return "conv groups must be multiples of 8 channels wide, odd kernel"
    if H % 8 or FFN % 8:
        return "dim_hidden / dim_ffn must be multiples of 8"
    return None


class NativeNBC2:
    """forward of one NBC2 module through the HIP building blocks; parameters are read from the module at every call (no copies)"""

    def __init__(self, net, lib: Lib):
        why = supported(net)
        if why is not None:
            raise NbssError(f"NBC2 native forward: {why}")
        self.net, self.lib = net, lib

    def _p(self, t: Optional[Tensor]):
        return ops._ptr(self.lib, t)  # (checks that the tensor lives where the library computes: HIP device — or host for the test emulator)

    def forward(self, x: Tensor) -> Tensor:
        """x [B,F,T,dim_input] (fp32 or bf16, HIP device) -> [B,F,T,dim_output] of the same dtype"""
        net, lib = self.net, self.lib
        B, F, T, Cin = x.shape
        if T > 256:
            raise NbssError(f"NBC2 native forward: {T} frames; the attention kernel keeps a sequence's K / V in LDS (<= 256 frames)")
        gs = net.sa_layers[0].norm2.group_size
        if F != gs:  # (the torch.nn module groups `group_size` consecutive sequences whatever F is; the kernel's groups are the utterances)
            raise NbssError(f"NBC2 native forward: {F} frequencies per utterance, GroupBatchNorm group_size {gs}")
        dt = NBSS_BF16 if x.dtype == torch.bfloat16 else NBSS_F32
        td = x.dtype if dt == NBSS_BF16 else torch.float32
        dev, nseq, N = x.device, B * F, B * F * T
        st = ops._stream(lib, x)
        H = net.encoder.out_channels
        blocks = list(net.sa_layers)
        FFN = blocks[0].linear1.out_features
        heads = blocks[0].self_attn.num_heads
        Cout = net.decoder.out_features
        # scratch for the re-laid weights of one launch (the largest of the network) and the LayerNorm statistics
        ks_e = net.encoder.kernel_size[0]
        g, ks = blocks[0].conv[1].groups, blocks[0].conv[1].kernel_size[0]
        need = [lib._dll.nbss_nb_ws_bytes(*a) for a in ((H, (Cin + 7) // 8 * 8, 1, ks_e), (3 * H, H, 1, 1), (H, H, 1, 1), (FFN, H, 1, 1), (FFN, FFN, g, ks),
                                                         (H, FFN, 1, 1), ((Cout + 7) // 8 * 8, H, 1, 1))]
        ws = torch.empty(max(need), dtype=torch.uint8, device=dev)
        stats = torch.empty(N, 2, dtype=torch.float32, device=dev)

        def f32(t):  # parameters as fp32 contiguous device tensors (they are: nn.Module parameters of an fp32 module)
            return t.detach().to(device=dev, dtype=torch.float32).contiguous()

        def conv(xin, cin, ldx, cout, groups, taps, w, b, res=None, act_in=0, act_out=0):
            y = torch.empty(nseq, T, cout, dtype=td, device=dev)
            w, b = f32(w), (f32(b) if b is not None else None)
            lib.call("nbss_nb_conv_t", dt, nseq, T, cin, ldx, cout, groups, taps, self._p(xin), self._p(w), self._p(b), self._p(y), self._p(res), act_in, act_out,
                     self._p(ws), st)
            return y

        def gbn(xin, mod, c, act):
            y = torch.empty_like(xin)
            w = f32(mod.weight.reshape(-1)) if mod.affine else None
            b = f32(mod.bias.reshape(-1)) if mod.affine else None
            lib.call("nbss_nb_group_batch_norm", dt, B, F, T, c, self._p(xin), self._p(w), self._p(b), C.c_float(mod.eps), act, self._p(y), st)
            return y

        # encoder: input columns padded to a multiple of 8 (zeros)
        Cin8 = (Cin + 7) // 8 * 8
        xin = torch.zeros(nseq, T, Cin8, dtype=td, device=dev)
        xin[..., :Cin] = x.reshape(nseq, T, Cin).to(td)
        h = conv(xin, Cin, Cin8, H, 1, ks_e, net.encoder.weight, net.encoder.bias)
        for b in blocks:
            u = torch.empty_like(h)
            lib.call("nbss_nb_layernorm", dt, N, H, self._p(h), self._p(f32(b.norm1.weight)), self._p(f32(b.norm1.bias)), self._p(u), self._p(stats), st)
            qkv = conv(u, H, H, 3 * H, 1, 1, b.self_attn.in_proj_weight, b.self_attn.in_proj_bias)
            o = torch.empty_like(h)
            lib.call("nbss_nb_attention_fwd", dt, nseq, T, H, heads, self._p(qkv), self._p(o), st)
            h = conv(o, H, H, H, 1, 1, b.self_attn.out_proj.weight, b.self_attn.out_proj.bias, res=h)
            v = gbn(h, b.norm2, H, 0)
            a = conv(v, H, H, FFN, 1, 1, b.linear1.weight, b.linear1.bias)
            c1 = conv(a, FFN, FFN, FFN, g, ks, b.conv[1].weight, b.conv[1].bias, act_in=1)
            c2 = conv(c1, FFN, FFN, FFN, g, ks, b.conv[3].weight, b.conv[3].bias, act_in=1)
            n3 = gbn(c2, b.conv[4], FFN, 1)
            c3 = conv(n3, FFN, FFN, FFN, g, ks, b.conv[6].weight, b.conv[6].bias)
            h = conv(c3, FFN, FFN, H, 1, 1, b.linear2.weight, b.linear2.bias, res=h, act_in=1)
        # decoder: output columns padded to a multiple of 8 (zero weight rows), sliced afterwards
        Co8 = (Cout + 7) // 8 * 8
        wd = torch.zeros(Co8, H, dtype=torch.float32, device=dev)
        wd[:Cout] = f32(net.decoder.weight)
        bd = torch.zeros(Co8, dtype=torch.float32, device=dev)
        bd[:Cout] = f32(net.decoder.bias)
        out = conv(h, H, H, Co8, 1, 1, wd, bd)
        return out[..., :Cout].reshape(B, F, T, Cout).to(x.dtype).contiguous()
