"""Native inference forward of the narrow-band conformer NBC (reference: models/arch/NBC.py:161-293: pre-norm blocks of Transformer-XL relative-position
attention and a convolutional feed-forward with GroupNorm) on the HIP device, sequenced from the `nbss_nb_*` building blocks of the C ABI:

  encoder   Conv1d(k, no padding: T -> T - k + 1 frames)                = a zero-padded tap-GEMM, rows k/2 .. of its output
  block     LayerNorm -> q | k | v maps -> nbss_nb_attention_relpos_fwd (P = pos_proj of the sinusoid table, one small GEMM per block)
            -> out_proj + residual;  LayerNorm -> linear1 -> SiLU -> 3 x (grouped conv -> GroupNorm(8) -> SiLU) -> linear2 + residual
  decoder   ConvTranspose1d(k)                                          = a zero-padded tap-GEMM over the frames shifted by one, taps flipped

Inference and training: `models.arch.NBC.NBC.forward` takes this path by default for every call on a HIP tensor the kernels support (NBSS_NBC_NATIVE=0
switches it off).  The reference trains NBC with dropout 0.1 inside the attention and the feed-forward (NBC.py:73-104,161-193): the attention dropout
goes through keep-bits both passes read (`_keep_bits`), the element-wise dropouts are device tensors.  `tests/test_nbc_native.py` runs both paths on the
emulator and on the device against the torch.nn module, `tests/test_nb_native_vs_reference.py` against numbers of the reference's own module."""
from __future__ import annotations

import math
import weakref
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


# ---- training (round 5): one autograd.Function whose backward walks the blocks in reverse over the nbss_nb_*_bwd building blocks, the relative-position
# attention through nbss_nb_attention_relpos_train / _bwd.  The reference's NBC trains with dropout 0.1 everywhere (NBC.py:83,168: not a constructor
# argument): the three element-wise dropouts of a block are torch ops on the device between the kernels (masks kept for backward), the attention dropout is
# a bit tensor [nseq][heads][T][ceil(T / 32)] drawn here with torch's generator and read by the forward AND the backward kernels. --------------------------------
def _param_list(net):
    ps = [net.encoder.weight, net.encoder.bias]
    for b in net.sa_layers:
        a = b.self_attn
        ps += [b.norm1.weight, b.norm1.bias, a.query_proj.weight, a.query_proj.bias, a.key_proj.weight, a.key_proj.bias, a.value_proj.weight, a.value_proj.bias,
               a.pos_proj.weight, a.u_bias, a.v_bias, a.out_proj.weight, a.out_proj.bias, b.norm2.weight, b.norm2.bias, b.linear1.weight, b.linear1.bias]
        for m in b.conv:
            if isinstance(m, (torch.nn.Conv1d, torch.nn.GroupNorm)):
                ps += [m.weight, m.bias]
        ps += [b.linear2.weight, b.linear2.bias]
    ps.append(net.decoder.weight)
    if net.decoder.bias is not None:
        ps.append(net.decoder.bias)
    return ps


def train_supported(net) -> Optional[str]:
    why = supported(net)
    if why is not None:
        return why
    if net.encoder.bias is None:
        return "encoder without bias"
    for b in net.sa_layers:
        a = b.self_attn
        if any(l.bias is None for l in (a.query_proj, a.key_proj, a.value_proj, a.out_proj, b.linear1, b.linear2)) or a.pos_proj.bias is not None:
            return "projections must have biases (pos_proj none)"
        for gn in (m for m in b.conv if isinstance(m, torch.nn.GroupNorm)):
            if (b.linear1.out_features // gn.num_groups) > 64:
                return "GroupNorm groups wider than 64 channels"
    if len(set(id(p) for p in _param_list(net))) != len(list(net.parameters())):
        return "parameters outside the native path"
    return None


class _NBCTrainFn(torch.autograd.Function):
    """out = NBC(x) with the gradients of every parameter from the HIP building blocks.  inputs: (runner, x, *parameters in _param_list order)"""

    @staticmethod
    def forward(ctx, runner, x, *params):
        out, saved = runner._forward_train(x)
        ops.graph_guard_save(ctx, runner, saved, params)
        return out

    @staticmethod
    def backward(ctx, dout):
        ops.graph_guard_check(ctx, "NBC native training")
        grads = ctx.runner._backward_train(ctx.saved, dout.contiguous())
        ctx.saved = None
        return (None, None, *grads)


def _keep_bits(shape, p: float, dev) -> Tensor:
    """attention-dropout keep-bits for [nseq, heads, T, T] probabilities: int32 words [nseq, heads, T, ceil(T / 32)], bit (j & 31) of word j >> 5 = (i, j) kept"""
    nseq, heads, T, _ = shape
    MW = (T + 31) // 32
    out = torch.empty(nseq, heads, T, MW, dtype=torch.int32, device=dev)
    wt = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)
    # one head at a time, bytes not int64 words: the transient is one float32 + one byte per probability of a head (the first version drew all heads at
    # once and packed through two int64 copies: ~10 GB of transients per layer at B 4 x F 257 x 248 frames x 8 heads)
    for h in range(heads):
        keep = (torch.rand(nseq, T, MW, 4, 8, device=dev) >= p).to(torch.uint8)
        out[:, h] = (keep * wt).sum(-1, dtype=torch.uint8).view(torch.int32).squeeze(-1)  # (little-endian: byte k of a word = bits 8k .. 8k + 7)
    return out


def forward_train(self, x: Tensor) -> Tensor:
    """training-mode forward with autograd: x [B,F,T,dim_input] -> [B,F,T,dim_output]; parameter gradients come from the HIP backward blocks"""
    why = train_supported(self.net)
    if why is not None:
        raise NbssError(f"NBC native training: {why}")
    return _NBCTrainFn.apply(self, x, *_param_list(self.net))


def _forward_train(self, x: Tensor):
    net, lib, p = self.net, self.lib, self._p
    B, F, T, Cin = x.shape
    K = net.encoder.kernel_size[0]
    Ti = T - K + 1
    if Ti < 1 or T > 256:
        raise NbssError(f"NBC native training: {T} frames (kernel {K}; the attention kernels keep a sequence and its offsets table in LDS: <= 256)")
    dt = NBSS_BF16 if x.dtype == torch.bfloat16 else NBSS_F32
    td = x.dtype if dt == NBSS_BF16 else torch.float32
    dev, nseq = x.device, B * F
    st = ops._stream(lib, x)
    H = net.encoder.out_channels
    blocks = list(net.sa_layers)
    FFN, heads, Cout = blocks[0].linear1.out_features, blocks[0].self_attn.num_heads, net.decoder.out_channels
    Cin8, Co8, K1 = (Cin + 7) // 8 * 8, (Cout + 7) // 8 * 8, K + 1
    cv0 = [m for m in blocks[0].conv if isinstance(m, torch.nn.Conv1d)][0]
    g, ks = cv0.groups, cv0.kernel_size[0]
    shapes = ((H, Cin8, 1, K1), (3 * H, H, 1, 1), (H, H, 1, 1), (FFN, H, 1, 1), (FFN, FFN, g, ks), (H, FFN, 1, 1), (Co8, H, 1, K1))
    ws = torch.empty(max(lib._dll.nbss_nb_bwd_ws_bytes(*a) for a in shapes), dtype=torch.uint8, device=dev)
    training = net.training
    keep = []  # converted copies stay alive until this call returns

    def f32(t):
        v = t.detach().to(device=dev, dtype=torch.float32).contiguous()
        keep.append(v)
        return v

    def conv(xin, rows_t, cin, ldx, cout, groups, taps, w, b, y2=False, n=nseq):
        y = torch.empty(n, rows_t, cout, dtype=td, device=dev)
        ys = torch.empty_like(y) if y2 else None
        w, b = f32(w), (f32(b) if b is not None else None)
        lib.call("nbss_nb_conv_t_train", dt, n, rows_t, cin, ldx, cout, groups, taps, p(xin), p(w), p(b), p(y), p(ys), None, p(ws), st)
        return (y, ys) if y2 else y

    def layernorm(h, mod, rows):
        u, stats = torch.empty_like(h), torch.empty(rows, 2, dtype=torch.float32, device=dev)
        lib.call("nbss_nb_layernorm", dt, rows, H, p(h), p(f32(mod.weight)), p(f32(mod.bias)), p(u), p(stats), st)
        return u, stats

    def dropout(t, mod):
        """-> (dropped tensor, mask scaled by 1 / (1 - p) in the stream dtype, or None)"""
        if not training or mod.p == 0.0:
            return t, None
        m = (torch.rand_like(t, dtype=torch.float32) >= mod.p).to(td) * (1.0 / (1.0 - mod.p))
        return t * m, m

    xin = torch.zeros(nseq, T, Cin8, dtype=td, device=dev)
    xin[..., :Cin] = x.reshape(nseq, T, Cin).to(td)
    wenc = torch.zeros(H, Cin8, K1, dtype=torch.float32, device=dev)
    wenc[:, :Cin, :K] = f32(net.encoder.weight)
    h = conv(xin, T, Cin8, Cin8, H, 1, K1, wenc, net.encoder.bias)[:, K // 2: K // 2 + Ti].contiguous()
    N = nseq * Ti
    per = []
    for b in blocks:
        a = b.self_attn
        u, st1 = layernorm(h, b.norm1, N)
        wqkv = torch.cat([f32(a.query_proj.weight), f32(a.key_proj.weight), f32(a.value_proj.weight)], 0)[..., None].contiguous()
        bqkv = torch.cat([f32(a.query_proj.bias), f32(a.key_proj.bias), f32(a.value_proj.bias)], 0)
        qkv = conv(u, Ti, H, H, 3 * H, 1, 1, wqkv, bqkv)
        pe = a.rel_pos.pe[0, a.rel_pos.zero_index - (Ti - 1): a.rel_pos.zero_index + Ti].to(device=dev, dtype=td).contiguous()[None]
        pos = conv(pe, 2 * Ti - 1, H, H, H, 1, 1, f32(a.pos_proj.weight)[..., None].contiguous(), None, n=1)
        pa = a.dropout.p if training else 0.0
        bits = _keep_bits((nseq, heads, Ti, Ti), pa, dev) if pa > 0 else None
        o = torch.empty_like(h)
        ub, vb = f32(a.u_bias), f32(a.v_bias)
        lib.call("nbss_nb_attention_relpos_train", dt, nseq, Ti, H, heads, p(qkv), p(pos), p(ub), p(vb), 1.0 / a.sqrt_dim, p(bits), 1.0 / (1.0 - pa), p(o), st)
        att = conv(o, Ti, H, H, H, 1, 1, f32(a.out_proj.weight)[..., None].contiguous(), a.out_proj.bias)
        att, m1 = dropout(att, b.dropout1)
        h1 = h + att
        v, st2 = layernorm(h1, b.norm2, N)
        a1, c = conv(v, Ti, H, H, FFN, 1, 1, f32(b.linear1.weight)[..., None].contiguous(), b.linear1.bias, y2=True)
        mods = list(b.conv)
        chain = []  # per conv step: (input c_prev, pre-norm z, GroupNorm stats)
        for i in range(0, len(mods), 3):
            cv, gn = mods[i], mods[i + 1]
            z = conv(c, Ti, FFN, FFN, FFN, cv.groups, cv.kernel_size[0], cv.weight, cv.bias)
            y = torch.empty_like(z)
            gst = torch.empty(nseq * gn.num_groups, 2, dtype=torch.float32, device=dev)
            lib.call("nbss_nb_group_norm_train", dt, nseq, Ti, FFN, gn.num_groups, p(z), p(f32(gn.weight)), p(f32(gn.bias)), 1, p(y), p(gst), st)
            chain.append((c, z, gst))
            c = y
        cd, md = dropout(c, b.dropout)
        f = conv(cd, Ti, FFN, FFN, H, 1, 1, f32(b.linear2.weight)[..., None].contiguous(), b.linear2.bias)
        f, m2 = dropout(f, b.dropout2)
        h2 = h1 + f
        per.append(dict(h=h, u=u, st1=st1, qkv=qkv, pe=pe, pos=pos, bits=bits, pa=pa, o=o, m1=m1, h1=h1, v=v, st2=st2, a1=a1, chain=chain, cd=cd, md=md, m2=m2))
        h = h2
    z = torch.zeros(nseq, T, H, dtype=td, device=dev)
    z[:, K // 2 - 1: K // 2 - 1 + Ti] = h
    wdec = torch.zeros(Co8, H, K1, dtype=torch.float32, device=dev)
    wdec[:Cout, :, :K] = f32(net.decoder.weight).permute(1, 0, 2).flip(-1)
    bdec = torch.zeros(Co8, dtype=torch.float32, device=dev)
    if net.decoder.bias is not None:
        bdec[:Cout] = f32(net.decoder.bias)
    out = conv(z, T, H, H, Co8, 1, K1, wdec, bdec)
    saved = dict(per=per, xin=xin, z=z, wenc=wenc, wdec=wdec, geo=(B, F, T, Ti, K, K1, Cin, Cin8, Cout, Co8, H, FFN, heads, g, ks, dt, td), ws=ws)
    return out[..., :Cout].reshape(B, F, T, Cout).to(x.dtype).contiguous(), saved


def _backward_train(self, sv, dout: Tensor):
    net, lib, p = self.net, self.lib, self._p
    B, F, T, Ti, K, K1, Cin, Cin8, Cout, Co8, H, FFN, heads, g, ks, dt, td = sv["geo"]
    dev, nseq, N = dout.device, B * F, B * F * Ti
    st = ops._stream(lib, dout)
    ws = sv["ws"]
    aws = torch.empty(lib._dll.nbss_nb_attention_relpos_bwd_ws_bytes(nseq, Ti, H, heads), dtype=torch.uint8, device=dev)
    blocks = list(net.sa_layers)
    keep = []

    def f32(t):
        v = t.detach().to(device=dev, dtype=torch.float32).contiguous()
        keep.append(v)
        return v

    def conv_bwd(xin, rows_t, cin, ldx, cout, groups, taps, w, dy, x_pre=None, need_dx=True, bias=True, n=nseq):
        """-> (dx or None, dw, db)"""
        dx = torch.empty(n, rows_t, ldx, dtype=td, device=dev) if need_dx else None
        dw = torch.zeros(cout * (cin // groups) * taps, dtype=torch.float32, device=dev)
        db = torch.zeros(cout, dtype=torch.float32, device=dev) if bias else None
        lib.call("nbss_nb_conv_t_bwd", dt, n, rows_t, cin, ldx, cout, groups, taps, p(xin), p(f32(w)), p(dy), p(x_pre), p(dx), p(dw), p(db), p(ws), st)
        return dx, dw, db

    def ln_bwd(xin, stats, mod, dy, dres):
        dx = torch.empty_like(xin)
        dg, db = torch.zeros(H, dtype=torch.float32, device=dev), torch.zeros(H, dtype=torch.float32, device=dev)
        lib.call("nbss_nb_layernorm_bwd", dt, N, H, p(xin), p(stats), p(f32(mod.weight)), p(dy), p(dres), p(dx), p(dg), p(db), st)
        return dx, dg, db

    # decoder (a transposed conv = the "same" conv of the zero-extended sequence with the taps flipped)
    d8 = torch.zeros(nseq, T, Co8, dtype=td, device=dev)
    d8[..., :Cout] = dout.reshape(nseq, T, Cout).to(td)
    dz, dwd, dbd = conv_bwd(sv["z"], T, H, H, Co8, 1, K1, sv["wdec"], d8)
    dh = dz[:, K // 2 - 1: K // 2 - 1 + Ti].contiguous()
    g_dec = [dwd.view(Co8, H, K1)[:Cout, :, :K].flip(-1).permute(1, 0, 2).contiguous()]
    if net.decoder.bias is not None:
        g_dec.append(dbd[:Cout])
    per_grads = []
    for b, s in zip(reversed(blocks), reversed(sv["per"])):
        a = b.self_attn
        # feed-forward branch: h2 = h1 + dropout2(linear2(dropout(chain(SiLU(linear1(LN2(h1)))))))
        df = dh * s["m2"] if s["m2"] is not None else dh
        dcd, dw2, db2 = conv_bwd(s["cd"], Ti, FFN, FFN, H, 1, 1, f32(b.linear2.weight)[..., None].contiguous(), df.contiguous())
        dc = dcd * s["md"] if s["md"] is not None else dcd
        mods = list(b.conv)
        chain_grads = []
        steps = [(mods[i], mods[i + 1]) for i in range(0, len(mods), 3)]
        for idx in range(len(steps) - 1, -1, -1):
            cv, gn = steps[idx]
            c_prev, z, gst = s["chain"][idx]
            dg, dbt = torch.zeros(FFN, dtype=torch.float32, device=dev), torch.zeros(FFN, dtype=torch.float32, device=dev)
            dzz = dc.contiguous()  # (ours alone: a conv_bwd output or the product with the dropout mask; the kernel works in place)
            lib.call("nbss_nb_group_norm_bwd", dt, nseq, Ti, FFN, gn.num_groups, p(z), p(gst), p(f32(gn.weight)), p(f32(gn.bias)), p(dzz), p(dg), p(dbt), st)
            # c_prev = SiLU(a1) for the first conv (x_pre: the gradient comes back multiplied by SiLU'(a1)), the previous step's output otherwise
            dc, dwc, dbc = conv_bwd(c_prev, Ti, FFN, FFN, FFN, cv.groups, cv.kernel_size[0], cv.weight, dzz, x_pre=s["a1"] if idx == 0 else None)
            chain_grads.append([dwc.reshape(cv.weight.shape), dbc, dg, dbt])
        dv, dw1, db1 = conv_bwd(s["v"], Ti, H, H, FFN, 1, 1, f32(b.linear1.weight)[..., None].contiguous(), dc)
        dh1, dg2, db2n = ln_bwd(s["h1"], s["st2"], b.norm2, dv, dh)
        # attention branch: h1 = h + dropout1(out_proj(attention(...)))
        da = dh1 * s["m1"] if s["m1"] is not None else dh1
        do, dwo, dbo = conv_bwd(s["o"], Ti, H, H, H, 1, 1, f32(a.out_proj.weight)[..., None].contiguous(), da.contiguous())
        dqkv = torch.empty_like(s["qkv"])
        dpos = torch.zeros(2 * Ti - 1, H, dtype=torch.float32, device=dev)
        dub, dvb = torch.zeros(H, dtype=torch.float32, device=dev), torch.zeros(H, dtype=torch.float32, device=dev)
        ub, vb = f32(a.u_bias), f32(a.v_bias)
        lib.call("nbss_nb_attention_relpos_bwd", dt, nseq, Ti, H, heads, p(s["qkv"]), p(s["pos"]), p(ub), p(vb), 1.0 / a.sqrt_dim, p(s["bits"]), 1.0 / (1.0 - s["pa"]),
                 p(do), p(dqkv), p(dpos), p(dub), p(dvb), p(aws), st)
        _, dwp, _ = conv_bwd(s["pe"], 2 * Ti - 1, H, H, H, 1, 1, f32(a.pos_proj.weight)[..., None].contiguous(), dpos.to(td)[None].contiguous(), need_dx=False, bias=False, n=1)
        wqkv = torch.cat([f32(a.query_proj.weight), f32(a.key_proj.weight), f32(a.value_proj.weight)], 0)[..., None].contiguous()
        du, dwi, dbi = conv_bwd(s["u"], Ti, H, H, 3 * H, 1, 1, wqkv, dqkv)
        dh, dg1, db1n = ln_bwd(s["h"], s["st1"], b.norm1, du, dh1)
        dwi = dwi.view(3, H, H)
        gl = [dg1, db1n, dwi[0], dbi[:H], dwi[1], dbi[H:2 * H], dwi[2], dbi[2 * H:], dwp.view(H, H), dub.view(a.u_bias.shape), dvb.view(a.v_bias.shape),
              dwo.view(H, H), dbo, dg2, db2n, dw1.view(FFN, H), db1]
        for cg in reversed(chain_grads):
            gl += cg
        gl += [dw2.view(H, FFN), db2]
        per_grads.append(gl)
    # encoder (no input gradient): the "valid" conv = rows K/2 .. of the zero-padded odd-kernel conv
    dfull = torch.zeros(nseq, T, H, dtype=td, device=dev)
    dfull[:, K // 2: K // 2 + Ti] = dh
    _, dwe, dbe = conv_bwd(sv["xin"], T, Cin8, Cin8, H, 1, K1, sv["wenc"], dfull, need_dx=False)
    grads = [dwe.view(H, Cin8, K1)[:, :Cin, :K].contiguous(), dbe]
    for gl in reversed(per_grads):
        grads += gl
    grads += g_dec
    return [gr.reshape(prm.shape).to(prm.dtype) for gr, prm in zip(grads, _param_list(net))]


NativeNBC.forward_train = forward_train
NativeNBC._forward_train = _forward_train
NativeNBC._backward_train = _backward_train
