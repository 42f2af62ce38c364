"""Native forward of the narrow-band conformer NBC2 (reference: models/arch/NBC2.py:152-289) on the HIP device, sequenced from the
geometry-generic building blocks of the C ABI (`nbss_nb_*`, csrc/gbwd.hip): the encoder Conv1d along time, per block LayerNorm ->
in_proj -> softmax(q k^T / sqrt(dh)) v per (sequence, head) -> out_proj + residual, GroupBatchNorm -> Linear -> SiLU -> grouped conv ->
SiLU -> grouped conv -> GroupBatchNorm -> SiLU -> grouped conv -> SiLU -> Linear + residual, and the decoder.  Every GEMM-shaped step is
one MFMA tap-GEMM launch (weights re-laid on the fly from the module's own fp32 parameters); activations stay in the
[B*F, T, C] layout of the reference.  Inference (`torch.no_grad()`): `NativeNBC2.forward`.  Training (round 4): `NativeNBC2.forward_train` —
one autograd.Function over the whole network whose backward walks the blocks with the `nbss_nb_*_bwd` entry points (transposed tap-GEMMs with
the SiLU' factor in their epilogue, the token-contraction weight gradients of csrc/wgrad.hip, LayerNorm / GroupBatchNorm / attention backward);
every parameter gradient comes from these kernels, torch contributes buffers and two residual adds per block.

`supported(net)` names what the kernels are built for: norms (LN, GBN, GBN) with per-frame GroupBatchNorm statistics, no dropout, head
width 24 or 48, channel counts that are multiples of 8 per conv group, sequences of at most 256 frames."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import weakref

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
        # (a weak reference: models/arch/* caches the runner in a WeakKeyDictionary keyed by the module — a strong reference from the value would keep
        #  every module that ever ran on the device, and its parameters, alive for the life of the process)
        self._net, self.lib = weakref.ref(net), lib

    @property
    def net(self):
        net = self._net()
        if net is None:
            raise NbssError("the module this native runner was built for has been freed")
        return net

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

        keep = []  # converted copies stay alive until this call returns: a temporary freed before its kernel is enqueued could be re-used by the next one

        def f32(t):  # parameters as fp32 contiguous device tensors (no copy for the fp32 parameters of an nn.Module)
            v = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(v)
            return v

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


class _NBC2TrainFn(torch.autograd.Function):
    """out = NBC2(x) with the gradients of every parameter from the HIP building blocks.  inputs: (runner, x, *parameters in runner.param_list order)"""

    @staticmethod
    def forward(ctx, runner, x, *params):
        out, saved = runner._forward_train(x)
        ops.graph_guard_save(ctx, runner, saved, params)
        return out

    @staticmethod
    def backward(ctx, dout):
        ops.graph_guard_check(ctx, "NBC2 native training")
        grads = ctx.runner._backward_train(ctx.saved, dout.contiguous())
        ctx.saved = None
        return (None, None, *grads)


def _param_list(net):
    ps = [net.encoder.weight, net.encoder.bias]
    for b in net.sa_layers:
        ps += [b.norm1.weight, b.norm1.bias, b.self_attn.in_proj_weight, b.self_attn.in_proj_bias, b.self_attn.out_proj.weight, b.self_attn.out_proj.bias,
               b.norm2.weight, b.norm2.bias, b.linear1.weight, b.linear1.bias, b.conv[1].weight, b.conv[1].bias, b.conv[3].weight, b.conv[3].bias,
               b.conv[4].weight, b.conv[4].bias, b.conv[6].weight, b.conv[6].bias, b.linear2.weight, b.linear2.bias]
    return ps + [net.decoder.weight, net.decoder.bias]


def _train_supported(net) -> Optional[str]:
    why = supported(net)
    if why is not None:
        return why
    for b in net.sa_layers:
        if not (b.norm2.affine and b.conv[4].affine):
            return "training path expects affine GroupBatchNorm"
    return None


def forward_train(self, x: Tensor) -> Tensor:
    """training-mode forward with autograd: x [B,F,T,dim_input] -> [B,F,T,dim_output]; parameter gradients come from the HIP backward blocks"""
    why = _train_supported(self.net)
    if why is not None:
        raise NbssError(f"NBC2 native training: {why}")
    return _NBC2TrainFn.apply(self, x, *_param_list(self.net))


def _forward_train(self, x: Tensor):
    net, lib = self.net, self.lib
    B, F, T, Cin = x.shape
    if T > 256:
        raise NbssError(f"NBC2 native training: {T} frames; the attention kernels keep a sequence's K / V in LDS (<= 256 frames)")
    gs = net.sa_layers[0].norm2.group_size
    if F != gs:
        raise NbssError(f"NBC2 native training: {F} frequencies per utterance, GroupBatchNorm group_size {gs}")
    dt = NBSS_BF16 if x.dtype == torch.bfloat16 else NBSS_F32
    td = x.dtype if dt == NBSS_BF16 else torch.float32
    dev, nseq, N = x.device, B * F, B * F * T
    st = ops._stream(lib, x)
    H = net.encoder.out_channels
    blocks = list(net.sa_layers)
    FFN, heads, Cout = blocks[0].linear1.out_features, blocks[0].self_attn.num_heads, net.decoder.out_features
    ks_e, g, ks = net.encoder.kernel_size[0], blocks[0].conv[1].groups, blocks[0].conv[1].kernel_size[0]
    Cin8, Co8 = (Cin + 7) // 8 * 8, (Cout + 7) // 8 * 8
    shapes = ((H, Cin8, 1, ks_e), (3 * H, H, 1, 1), (H, H, 1, 1), (FFN, H, 1, 1), (FFN, FFN, g, ks), (H, FFN, 1, 1), (Co8, H, 1, 1))
    ws = torch.empty(max(lib._dll.nbss_nb_bwd_ws_bytes(*a) for a in shapes), dtype=torch.uint8, device=dev)
    p = self._p

    keep = []  # converted copies stay alive until this call returns: a temporary freed before its kernel is enqueued could be re-used by the next one

    def f32(t):  # parameters as fp32 contiguous device tensors (no copy for the fp32 parameters of an nn.Module)
        v = t.detach().to(device=dev, dtype=torch.float32).contiguous()
        keep.append(v)
        return v

    def conv(xin, cin, ldx, cout, groups, taps, w, b, res=None, y2=False):
        y = torch.empty(nseq, T, cout, dtype=td, device=dev)
        ys = torch.empty_like(y) if y2 else None
        lib.call("nbss_nb_conv_t_train", dt, nseq, T, cin, ldx, cout, groups, taps, p(xin), p(f32(w)), p(f32(b)) if b is not None else None, p(y), p(ys), p(res), p(ws), st)
        return (y, ys) if y2 else y

    def gbn(xin, mod, c, act):
        y = torch.empty_like(xin)
        lib.call("nbss_nb_group_batch_norm", dt, B, F, T, c, p(xin), p(f32(mod.weight.reshape(-1))), p(f32(mod.bias.reshape(-1))), C.c_float(mod.eps), act, p(y), st)
        return y

    xin = torch.zeros(nseq, T, Cin8, dtype=td, device=dev)
    xin[..., :Cin] = x.reshape(nseq, T, Cin).to(td)
    h = conv(xin, Cin, Cin8, H, 1, ks_e, net.encoder.weight, net.encoder.bias)
    per = []
    for b in blocks:
        u, stats = torch.empty_like(h), torch.empty(N, 2, dtype=torch.float32, device=dev)
        lib.call("nbss_nb_layernorm", dt, N, H, p(h), p(f32(b.norm1.weight)), p(f32(b.norm1.bias)), p(u), p(stats), st)
        qkv = conv(u, H, H, 3 * H, 1, 1, b.self_attn.in_proj_weight, b.self_attn.in_proj_bias)
        o = torch.empty_like(h)
        lib.call("nbss_nb_attention_fwd", dt, nseq, T, H, heads, p(qkv), p(o), st)
        h1 = conv(o, H, H, H, 1, 1, b.self_attn.out_proj.weight, b.self_attn.out_proj.bias, res=h)
        v = gbn(h1, b.norm2, H, 0)
        a, sa = conv(v, H, H, FFN, 1, 1, b.linear1.weight, b.linear1.bias, y2=True)
        c1, sc1 = conv(sa, FFN, FFN, FFN, g, ks, b.conv[1].weight, b.conv[1].bias, y2=True)
        c2 = conv(sc1, FFN, FFN, FFN, g, ks, b.conv[3].weight, b.conv[3].bias)
        n3 = gbn(c2, b.conv[4], FFN, 1)
        c3, sc3 = conv(n3, FFN, FFN, FFN, g, ks, b.conv[6].weight, b.conv[6].bias, y2=True)
        h2 = conv(sc3, FFN, FFN, H, 1, 1, b.linear2.weight, b.linear2.bias, res=h1)
        per.append(dict(h=h, u=u, stats=stats, qkv=qkv, o=o, h1=h1, v=v, a=a, sa=sa, c1=c1, sc1=sc1, c2=c2, n3=n3, c3=c3, sc3=sc3))
        h = h2
    wd = torch.zeros(Co8, H, dtype=torch.float32, device=dev)
    wd[:Cout] = f32(net.decoder.weight)
    bd = torch.zeros(Co8, dtype=torch.float32, device=dev)
    bd[:Cout] = f32(net.decoder.bias)
    out = conv(h, H, H, Co8, 1, 1, wd, bd)
    saved = dict(per=per, xin=xin, hL=h, wd=wd, geo=(B, F, T, Cin, Cin8, Cout, Co8, H, FFN, heads, g, ks, ks_e, dt, td), ws=ws, out_dtype=x.dtype)
    return out[..., :Cout].reshape(B, F, T, Cout).to(x.dtype).contiguous(), saved


def _backward_train(self, sv, dout: Tensor):
    net, lib, p = self.net, self.lib, self._p
    B, F, T, Cin, Cin8, Cout, Co8, H, FFN, heads, g, ks, ks_e, dt, td = sv["geo"]
    dev, nseq, N = dout.device, B * F, B * F * T
    st = ops._stream(lib, dout)
    ws = sv["ws"]
    aws = torch.empty(lib._dll.nbss_nb_attention_bwd_ws_bytes(dt, nseq, T, H, heads), dtype=torch.uint8, device=dev)
    blocks = list(net.sa_layers)

    keep = []  # converted copies stay alive until this call returns: a temporary freed before its kernel is enqueued could be re-used by the next one

    def f32(t):  # parameters as fp32 contiguous device tensors (no copy for the fp32 parameters of an nn.Module)
        v = t.detach().to(device=dev, dtype=torch.float32).contiguous()
        keep.append(v)
        return v

    def zeros_like_param(t):
        return torch.zeros(t.numel(), dtype=torch.float32, device=dev)

    def conv_bwd(xin, cin, ldx, cout, groups, taps, w, dy, x_pre=None, need_dx=True, bias=True):
        """-> (dx or None, dw, db)"""
        dx = torch.empty(nseq, T, ldx, dtype=td, device=dev) if need_dx else None
        dw = torch.zeros(cout * (cin // groups) * taps, dtype=torch.float32, device=dev)
        db = torch.zeros(cout, dtype=torch.float32, device=dev) if bias else None
        lib.call("nbss_nb_conv_t_bwd", dt, nseq, T, cin, ldx, cout, groups, taps, p(xin), p(f32(w)), p(dy), p(x_pre), p(dx), p(dw), p(db), p(ws), st)
        return dx, dw, db

    def gbn_bwd(xin, mod, c, act, dy):
        dx = torch.empty_like(xin)
        dg, dbt = torch.zeros(c, dtype=torch.float32, device=dev), torch.zeros(c, dtype=torch.float32, device=dev)
        lib.call("nbss_nb_group_batch_norm_bwd", dt, B, F, T, c, p(xin), p(f32(mod.weight.reshape(-1))), p(f32(mod.bias.reshape(-1))), C.c_float(mod.eps), act, p(dy), p(dx),
                 p(dg), p(dbt), st)
        return dx, dg, dbt

    # decoder
    d8 = torch.zeros(nseq, T, Co8, dtype=td, device=dev)
    d8[..., :Cout] = dout.reshape(nseq, T, Cout).to(td)
    dh, dwd, dbd = conv_bwd(sv["hL"], H, H, Co8, 1, 1, sv["wd"], d8)
    g_dec = [dwd.reshape(Co8, H)[:Cout].reshape(net.decoder.weight.shape), dbd[:Cout]]
    per_grads = []
    for b, s in zip(reversed(blocks), reversed(sv["per"])):
        # feed-forward branch: h2 = linear2(SiLU(c3)) + h1
        dc3, dw2, db2 = conv_bwd(s["sc3"], FFN, FFN, H, 1, 1, b.linear2.weight, dh, x_pre=s["c3"])
        dn3, dwc3, dbc3 = conv_bwd(s["n3"], FFN, FFN, FFN, g, ks, b.conv[6].weight, dc3)
        dc2, dgn3, dbn3 = gbn_bwd(s["c2"], b.conv[4], FFN, 1, dn3)
        dc1, dwc2, dbc2 = conv_bwd(s["sc1"], FFN, FFN, FFN, g, ks, b.conv[3].weight, dc2, x_pre=s["c1"])
        da, dwc1, dbc1 = conv_bwd(s["sa"], FFN, FFN, FFN, g, ks, b.conv[1].weight, dc1, x_pre=s["a"])
        dv, dw1, db1 = conv_bwd(s["v"], H, H, FFN, 1, 1, b.linear1.weight, da)
        dh1b, dgn2, dbn2 = gbn_bwd(s["h1"], b.norm2, H, 0, dv)
        dh1 = dh + dh1b  # residual: h2 = h1 + ffn(h1)
        # attention branch: h1 = out_proj(attn(in_proj(LN(h)))) + h
        do, dwo, dbo = conv_bwd(s["o"], H, H, H, 1, 1, b.self_attn.out_proj.weight, dh1)
        dqkv = torch.empty_like(s["qkv"])
        lib.call("nbss_nb_attention_bwd", dt, nseq, T, H, heads, p(s["qkv"]), p(do), p(dqkv), p(aws), st)
        du, dwi, dbi = conv_bwd(s["u"], H, H, 3 * H, 1, 1, b.self_attn.in_proj_weight, dqkv)
        dhn = torch.empty_like(dh1)
        dg1, db1n = torch.zeros(H, dtype=torch.float32, device=dev), torch.zeros(H, dtype=torch.float32, device=dev)
        lib.call("nbss_nb_layernorm_bwd", dt, N, H, p(s["h"]), p(s["stats"]), p(f32(b.norm1.weight)), p(du), p(dh1), p(dhn), p(dg1), p(db1n), st)
        dh = dhn
        per_grads.append([dg1, db1n, dwi.reshape(b.self_attn.in_proj_weight.shape), dbi, dwo.reshape(b.self_attn.out_proj.weight.shape), dbo,
                          dgn2.reshape(b.norm2.weight.shape), dbn2.reshape(b.norm2.bias.shape), dw1.reshape(b.linear1.weight.shape), db1,
                          dwc1.reshape(b.conv[1].weight.shape), dbc1, dwc2.reshape(b.conv[3].weight.shape), dbc2,
                          dgn3.reshape(b.conv[4].weight.shape), dbn3.reshape(b.conv[4].bias.shape), dwc3.reshape(b.conv[6].weight.shape), dbc3,
                          dw2.reshape(b.linear2.weight.shape), db2])
    # encoder (no input gradient)
    _, dwe, dbe = conv_bwd(sv["xin"], Cin, Cin8, H, 1, ks_e, net.encoder.weight, dh, need_dx=False)
    grads = [dwe.reshape(net.encoder.weight.shape), dbe]
    for gl in reversed(per_grads):
        grads += gl
    grads += g_dec
    return [gr.to(prm.dtype) for gr, prm in zip(grads, _param_list(net))]


NativeNBC2.forward_train = forward_train
NativeNBC2._forward_train = _forward_train
NativeNBC2._backward_train = _backward_train
