"""Native inference forward of the narrow-band conformer NBC (reference: models/arch/NBC.py:161-293: pre-norm blocks of Transformer-XL relative-position
attention and a convolutional feed-forward with GroupNorm) on the HIP device, sequenced from the `nbss_nb_*` building blocks of the C ABI:

  encoder   Conv1d(k, no padding: T -> T - k + 1 frames)                = a zero-padded tap-GEMM, rows k/2 .. of its output
  block     LayerNorm -> q | k | v maps -> nbss_nb_attention_relpos_fwd (P = pos_proj of the sinusoid table, one small GEMM per block)
            -> out_proj + residual;  LayerNorm -> linear1 -> SiLU -> 3 x (grouped conv -> GroupNorm(8) -> SiLU) -> linear2 + residual
  decoder   ConvTranspose1d(k)                                          = a zero-padded tap-GEMM over the frames shifted by one, taps flipped

Inference only: the reference trains NBC with dropout 0.1 inside the attention and the feed-forward (NBC.py:73-104,161-193), which these kernels do
not draw; `models.arch.NBC.NBC.forward` takes this path for eval-mode / no-grad calls on a HIP tensor when NBSS_NBC_NATIVE=1 (opt-in until its first
run on the device: round 4 ended before one), `tests/test_nbc_native.py` runs it on the emulator against the torch.nn module."""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor

from . import ops
from ._lib import NBSS_BF16, NBSS_F32, Lib, NbssError


def supported(net) -> Optional[str]:
    """None when `net` (models.arch.NBC.NBC) can run through the native forward, else the reason"""
    blocks = list(net.sa_layers)
    if not blocks:
        return "no layers"
    H = net.encoder.out_channels
    if net.encoder.groups != 1 or net.encoder.stride[0] != 1 or net.decoder.stride[0] != 1 or net.decoder.kernel_size != net.encoder.kernel_size:
        return "encoder / decoder must be dense stride-1 (transposed) convolutions of one kernel size"
    if net.encoder.kernel_size[0] != 4:
        return "encoder kernel size must be 4"
    for b in blocks:
        if not b.norm_first:
            return "norm_first = False"
        a = b.self_attn
        if H // a.num_heads not in (24, 48):
            return "attention head width must be 24 or 48"
        convs = [m for m in b.conv if isinstance(m, torch.nn.Conv1d)]
        gns = [m for m in b.conv if isinstance(m, torch.nn.GroupNorm)]
        if len(convs) != len(gns) or len(list(b.conv)) != 3 * len(convs):
            return "the feed-forward must be (conv, GroupNorm, SiLU) x n"
        for cv in convs:
            if cv.kernel_size[0] % 2 == 0 or (cv.in_channels // cv.groups) % 8 or cv.bias is None:
                return "grouped convs need an odd kernel, a bias and groups that are multiples of 8 channels wide"
            if cv.in_channels != b.linear1.out_features or cv.out_channels != b.linear1.out_features:
                return "the feed-forward convs must be ffn_size wide"
        for gn in gns:
            if abs(gn.eps - 1e-5) > 1e-12 or not gn.affine:
                return "GroupNorm must be affine with eps 1e-5"
    if H % 8 or blocks[0].linear1.out_features % 8:
        return "hidden_size / ffn_size must be multiples of 8"
    return None


class NativeNBC:
    """inference forward of one NBC module through the HIP building blocks; parameters are read from the module at every call (no copies)"""

    def __init__(self, net, lib: Lib):
        why = supported(net)
        if why is not None:
            raise NbssError(f"NBC native forward: {why}")
        self.net, self.lib = net, lib

    def _p(self, t: Optional[Tensor]):
        return ops._ptr(self.lib, t)

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        """x [B,F,T,dim_input] (fp32 or bf16) -> [B,F,T,dim_output] of the same dtype (dropout inactive: eval semantics)"""
        net, lib = self.net, self.lib
        B, F, T, Cin = x.shape
        K = net.encoder.kernel_size[0]
        Ti = T - K + 1  # frames inside the network
        if Ti < 1 or T > 256:
            raise NbssError(f"NBC native forward: {T} frames (kernel {K}; the attention kernel keeps a sequence's K / V / offsets table in LDS: <= 256)")
        dt = NBSS_BF16 if x.dtype == torch.bfloat16 else NBSS_F32
        td = x.dtype if dt == NBSS_BF16 else torch.float32
        dev, nseq = x.device, B * F
        st = ops._stream(lib, x)
        H = net.encoder.out_channels
        blocks = list(net.sa_layers)
        FFN = blocks[0].linear1.out_features
        heads = blocks[0].self_attn.num_heads
        Cout = net.decoder.out_channels
        Cin8, Co8 = (Cin + 7) // 8 * 8, (Cout + 7) // 8 * 8
        cv0 = [m for m in blocks[0].conv if isinstance(m, torch.nn.Conv1d)][0]
        K1 = K + 1  # (the building block takes odd kernels: one more tap with zero weights, same centre K/2)
        need = [lib._dll.nbss_nb_ws_bytes(*a) for a in ((H, Cin8, 1, K1), (3 * H, H, 1, 1), (H, H, 1, 1), (FFN, H, 1, 1), (FFN, FFN, cv0.groups, cv0.kernel_size[0]),
                                                         (H, FFN, 1, 1), (Co8, H, 1, K1))]
        ws = torch.empty(max(need), dtype=torch.uint8, device=dev)

        keep = []  # converted copies stay alive until this call returns: a temporary freed before its kernel is enqueued could be re-used by the next one

        def f32(t):  # parameters as fp32 contiguous device tensors (no copy for the fp32 parameters of an nn.Module)
            v = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(v)
            return v

        def conv(xin, rows_t, cin, ldx, cout, groups, taps, w, b, res=None, act_in=0, act_out=0, n=nseq):
            y = torch.empty(n, rows_t, cout, dtype=td, device=dev)
            w, b = f32(w), (f32(b) if b is not None else None)
            lib.call("nbss_nb_conv_t", dt, n, rows_t, cin, ldx, cout, groups, taps, self._p(xin), self._p(w), self._p(b), self._p(y), self._p(res), act_in, act_out,
                     self._p(ws), st)
            return y

        def layernorm(h, mod, rows):
            u, stats = torch.empty_like(h), torch.empty(rows, 2, dtype=torch.float32, device=dev)
            lib.call("nbss_nb_layernorm", dt, rows, H, self._p(h), self._p(f32(mod.weight)), self._p(f32(mod.bias)), self._p(u), self._p(stats), st)
            return u

        # encoder: y[t'] = sum_k x[t' + k] w[k] (t' < T - K + 1) = rows K/2 .. of the zero-padded ("same", centre K/2) conv over the T frames
        xin = torch.zeros(nseq, T, Cin8, dtype=td, device=dev)
        xin[..., :Cin] = x.reshape(nseq, T, Cin).to(td)
        wenc = torch.zeros(H, Cin8, K1, dtype=torch.float32, device=dev)
        wenc[:, :Cin, :K] = f32(net.encoder.weight)
        h = conv(xin, T, Cin8, Cin8, H, 1, K1, wenc, net.encoder.bias)[:, K // 2: K // 2 + Ti].contiguous()
        N = nseq * Ti
        for b in blocks:
            a = b.self_attn
            u = layernorm(h, b.norm1, N)
            wqkv = torch.cat([f32(a.query_proj.weight), f32(a.key_proj.weight), f32(a.value_proj.weight)], 0)[..., None]
            bqkv = torch.cat([f32(a.query_proj.bias), f32(a.key_proj.bias), f32(a.value_proj.bias)], 0)
            qkv = conv(u, Ti, H, H, 3 * H, 1, 1, wqkv, bqkv)
            # P = pos_proj(sinusoid rows for the offsets -(Ti - 1) .. Ti - 1): one [2 Ti - 1][H] x [H][H] map (a single "sequence")
            pe = a.rel_pos.pe[0, a.rel_pos.zero_index - (Ti - 1): a.rel_pos.zero_index + Ti].to(device=dev, dtype=td).contiguous()[None]
            pos = conv(pe, 2 * Ti - 1, H, H, H, 1, 1, f32(a.pos_proj.weight)[..., None], None, n=1)
            o = torch.empty_like(h)
            lib.call("nbss_nb_attention_relpos_fwd", dt, nseq, Ti, H, heads, self._p(qkv), self._p(pos), self._p(f32(a.u_bias)), self._p(f32(a.v_bias)),
                     1.0 / a.sqrt_dim, self._p(o), st)
            h = conv(o, Ti, H, H, H, 1, 1, f32(a.out_proj.weight)[..., None], a.out_proj.bias, res=h)
            v = layernorm(h, b.norm2, N)
            c = conv(v, Ti, H, H, FFN, 1, 1, f32(b.linear1.weight)[..., None], b.linear1.bias, act_out=1)
            mods = list(b.conv)
            for i in range(0, len(mods), 3):
                cv, gn = mods[i], mods[i + 1]
                c = conv(c, Ti, FFN, FFN, FFN, cv.groups, cv.kernel_size[0], cv.weight, cv.bias)
                y = torch.empty_like(c)
                lib.call("nbss_nb_group_norm", dt, nseq, Ti, FFN, gn.num_groups, self._p(c), self._p(f32(gn.weight)), self._p(f32(gn.bias)), 1, self._p(y), st)
                c = y
            h = conv(c, Ti, FFN, FFN, H, 1, 1, f32(b.linear2.weight)[..., None], b.linear2.bias, res=h)
        # decoder: y[t] = sum_k h[t - k] w[:, :, k] over T = Ti + K - 1 frames = the "same" conv (centre K/2) of z, z[j] = h[j - (K/2 - 1)], with the taps
        # flipped: offset d = tap - K/2 reads z[t + d] = h[t - k] for k = K/2 - 1 - d... (K = 4: k = 3 - tap)
        z = torch.zeros(nseq, T, H, dtype=td, device=dev)
        z[:, K // 2 - 1: K // 2 - 1 + Ti] = h
        wdec = torch.zeros(Co8, H, K1, dtype=torch.float32, device=dev)
        wdec[:Cout, :, :K] = f32(net.decoder.weight).permute(1, 0, 2).flip(-1)  # ConvTranspose1d weight is [in][out][k]
        bdec = torch.zeros(Co8, dtype=torch.float32, device=dev)
        if net.decoder.bias is not None:
            bdec[:Cout] = f32(net.decoder.bias)
        out = conv(z, T, H, H, Co8, 1, K1, wdec, bdec)
        return out[..., :Cout].reshape(B, F, T, Cout).to(x.dtype).contiguous()
