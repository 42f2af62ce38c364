"""NB-BLSTM (models/arch/blstm2_fc1.py; reference blstm2_fc1.py:45-68) on the HIP building blocks: per bidirectional layer ONE dense map for the input part of the
gates (nbss_nb_conv_t), ONE persistent launch for the recurrences of both directions over all frames (nbss_nb_blstm_fwd: gates GEMM on MFMA, h in LDS, c in
registers), and in training the reverse walk (nbss_nb_blstm_bwd) followed by dense contractions for the weight / bias / input gradients (nbss_nb_conv_t_bwd).
The module keeps its nn.LSTM parameters (state_dict keys unchanged); they are read at every call."""
from typing import Optional

import weakref

import torch
from torch import Tensor

from . import ops
from ._lib import NBSS_BF16, NBSS_F32, Lib, NbssError


def supported(net) -> Optional[str]:
    """None when `net` (models.arch.blstm2_fc1.BLSTM2_FC1) can run through the native path, else the reason"""
    for rnn in (net.blstm1, net.blstm2):
        if rnn.hidden_size not in (128, 256):
            return f"hidden size {rnn.hidden_size} (the recurrence kernels are built for 128 and 256)"
        if rnn.num_layers != 1 or not rnn.bidirectional or not rnn.batch_first or not rnn.bias or rnn.proj_size != 0:
            return "LSTM layers must be single bidirectional batch-first layers with biases"
    if net.dropout:
        return "dropout between the layers"
    if net.activation_func is not None:
        return "output activation"
    return None


def _param_list(net):
    ps = []
    for rnn in (net.blstm1, net.blstm2):
        for sfx in ("", "_reverse"):
            ps += [getattr(rnn, f"weight_ih_l0{sfx}"), getattr(rnn, f"weight_hh_l0{sfx}"), getattr(rnn, f"bias_ih_l0{sfx}"), getattr(rnn, f"bias_hh_l0{sfx}")]
    return ps + [net.linear.weight, net.linear.bias]


class _BLSTMTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, x, *params):
        out, saved = runner._run(x, train=True)
        ops.graph_guard_save(ctx, runner, saved, params)
        return out

    @staticmethod
    def backward(ctx, dout):
        ops.graph_guard_check(ctx, "NB-BLSTM native training")
        grads = ctx.runner._backward(ctx.saved, dout.contiguous())
        ctx.saved = None
        return (None, None, *grads)


class NativeBLSTM:
    def __init__(self, net, lib: Lib):
        why = supported(net)
        if why is not None:
            raise NbssError(f"NB-BLSTM native path: {why}")
        # (a weak reference: models/arch/* caches the runner in a WeakKeyDictionary keyed by the module — a strong reference from the value would keep
        #  every module that ever ran on the device, and its parameters, alive for the life of the process)
        self._net, self.lib = weakref.ref(net), lib

    @property
    def net(self):
        net = self._net()
        if net is None:
            raise NbssError("the module this native runner was built for has been freed")
        return net

    def _p(self, t):
        return ops._ptr(self.lib, t)

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        return self._run(x, train=False)[0]

    def forward_train(self, x: Tensor) -> Tensor:
        return _BLSTMTrainFn.apply(self, x, *_param_list(self.net))

    def _run(self, x: Tensor, train: bool):
        net, lib, p = self.net, self.lib, self._p
        B, F, T, Cin = x.shape
        dt = NBSS_BF16 if x.dtype == torch.bfloat16 else NBSS_F32
        td = x.dtype if dt == NBSS_BF16 else torch.float32
        dev, n = x.device, B * F
        st = ops._stream(lib, x)
        keep = []

        def f32(t):
            v = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(v)
            return v

        h = x.reshape(n, T, Cin).to(td)
        layers = []
        for rnn in (net.blstm1, net.blstm2):
            HD, I = rnn.hidden_size, h.shape[-1]
            I8 = (I + 7) // 8 * 8
            if I8 != I:
                hp = torch.zeros(n, T, I8, dtype=td, device=dev)
                hp[..., :I] = h
                h = hp
            h = h.contiguous()
            wih = torch.zeros(8 * HD, I8, 1, dtype=torch.float32, device=dev)
            wih[:4 * HD, :I, 0] = f32(rnn.weight_ih_l0)
            wih[4 * HD:, :I, 0] = f32(rnn.weight_ih_l0_reverse)
            bias = torch.cat([f32(rnn.bias_ih_l0) + f32(rnn.bias_hh_l0), f32(rnn.bias_ih_l0_reverse) + f32(rnn.bias_hh_l0_reverse)])
            ws = torch.empty(max(lib._dll.nbss_nb_bwd_ws_bytes(8 * HD, I8, 1, 1), lib._dll.nbss_nb_blstm_ws_bytes(dt, HD)), dtype=torch.uint8, device=dev)
            gx = torch.empty(n, T, 8 * HD, dtype=td, device=dev)
            lib.call("nbss_nb_conv_t_train", dt, n, T, I8, I8, 8 * HD, 1, 1, p(h), p(wih), p(bias), p(gx), None, None, p(ws), st)
            y = torch.empty(n, T, 2 * HD, dtype=td, device=dev)
            save = torch.empty(2, n, T, 5 * HD, dtype=td, device=dev) if train else None
            whh0, whh1 = f32(rnn.weight_hh_l0), f32(rnn.weight_hh_l0_reverse)
            lib.call("nbss_nb_blstm_fwd", dt, n, T, HD, 8 * HD, p(gx), p(whh0), p(whh1), p(y), p(save), p(ws), st)
            layers.append(dict(x=h, I=I, I8=I8, HD=HD, wih=wih, y=y, save=save))
            h = y
        Cout = net.linear.out_features
        Co8, K2 = (Cout + 7) // 8 * 8, h.shape[-1]
        wl = torch.zeros(Co8, K2, 1, dtype=torch.float32, device=dev)
        wl[:Cout, :, 0] = f32(net.linear.weight)
        bl = torch.zeros(Co8, dtype=torch.float32, device=dev)
        bl[:Cout] = f32(net.linear.bias)
        ws = torch.empty(lib._dll.nbss_nb_bwd_ws_bytes(Co8, K2, 1, 1), dtype=torch.uint8, device=dev)
        out = torch.empty(n, T, Co8, dtype=td, device=dev)
        lib.call("nbss_nb_conv_t_train", dt, n, T, K2, K2, Co8, 1, 1, p(h), p(wl), p(bl), p(out), None, None, p(ws), st)
        saved = dict(layers=layers, wl=wl, geo=(B, F, T, n, Cout, Co8, dt, td)) if train else None
        return out[..., :Cout].reshape(B, F, T, Cout).to(x.dtype).contiguous(), saved

    def _backward(self, sv, dout: Tensor):
        net, lib, p = self.net, self.lib, self._p
        B, F, T, n, Cout, Co8, dt, td = sv["geo"]
        dev = dout.device
        st = ops._stream(lib, dout)
        keep = []

        def f32(t):
            v = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(v)
            return v

        def dense_bwd(xin, cin, cout, w, dy, need_dx=True):
            """gradients of y = x w^T + b over all (sequence, frame) rows -> (dx or None, dw [cout][cin], db [cout])"""
            ws = torch.empty(lib._dll.nbss_nb_bwd_ws_bytes(cout, cin, 1, 1), dtype=torch.uint8, device=dev)
            dx = torch.empty(n, T, cin, dtype=td, device=dev) if need_dx else None
            dw = torch.zeros(cout * cin, dtype=torch.float32, device=dev)
            db = torch.zeros(cout, dtype=torch.float32, device=dev)
            lib.call("nbss_nb_conv_t_bwd", dt, n, T, cin, cin, cout, 1, 1, p(xin), p(w), p(dy), None, p(dx), p(dw), p(db), p(ws), st)
            return dx, dw.view(cout, cin), db

        L1, L2 = sv["layers"]
        d8 = torch.zeros(n, T, Co8, dtype=td, device=dev)
        d8[..., :Cout] = dout.reshape(n, T, Cout).to(td)
        dy, dwl, dbl = dense_bwd(L2["y"], L2["y"].shape[-1], Co8, sv["wl"], d8)
        grads_rev = [[dwl[:Cout], dbl[:Cout]]]
        for L, rnn, need_dx in ((L2, net.blstm2, True), (L1, net.blstm1, False)):
            HD, I, I8 = L["HD"], L["I"], L["I8"]
            ws = torch.empty(lib._dll.nbss_nb_blstm_ws_bytes(dt, HD), dtype=torch.uint8, device=dev)
            dg = torch.empty(n, T, 8 * HD, dtype=td, device=dev)
            whh0, whh1 = f32(rnn.weight_hh_l0), f32(rnn.weight_hh_l0_reverse)
            dyc = dy.contiguous()  # (named: a temporary would be returned to the allocator before the call that reads it is made)
            lib.call("nbss_nb_blstm_bwd", dt, n, T, HD, p(dyc), p(L["save"]), p(whh0), p(whh1), p(dg), p(ws), st)
            dx, dwih, dbih = dense_bwd(L["x"], I8, 8 * HD, L["wih"], dg, need_dx=need_dx)
            # recurrent weights: dW_hh = sum over (sequence, frame) of dG_t^T h_{t-1}; h_{t-1} = the direction's output one frame earlier in ITS order
            y = L["y"]
            gl = []
            for d in range(2):
                hp = torch.zeros(n, T, HD, dtype=td, device=dev)
                if d == 0:
                    hp[:, 1:] = y[:, :-1, :HD]
                else:
                    hp[:, :-1] = y[:, 1:, HD:]
                _, dwhh, _ = dense_bwd(hp, HD, 4 * HD, (whh0, whh1)[d][..., None].contiguous(), dg[..., 4 * HD * d: 4 * HD * (d + 1)].contiguous(), need_dx=False)
                b = dbih[4 * HD * d: 4 * HD * (d + 1)]
                gl += [dwih[4 * HD * d: 4 * HD * (d + 1), :I], dwhh, b, b.clone()]
            grads_rev.append(gl)
            dy = dx[..., :I] if need_dx else None
        grads = grads_rev[2] + grads_rev[1] + grads_rev[0]
        return [g.reshape(prm.shape).to(prm.dtype) for g, prm in zip(grads, _param_list(net))]
