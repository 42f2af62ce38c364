"""Native streaming step of an OnlineSpatialNet (BASELINE config 5): the HIP kernels of csrc/online.hip for the
narrow-band halves (causal encoder, recurrent multi-scale retention, causal T-ConvFFN with its per-frame cross-frequency GroupNorm) and the
existing cross-band kernels (nbss_fconv_fwd / nbss_full_fwd) and decoder, on a chunk of C frames with all state in pre-allocated device
buffers — a fixed launch sequence that is captured once into a HIP graph and replayed per chunk.

`NativeOnlineStreamer(net, batch, chunk)` has the interface of models.arch.OnlineSpatialNet.OnlineStreamer (step / reset / graph) and serves
the geometry the kernels are built for: attention 'ret(F, share_qk | not_share_qk)' with value factor 2 and no rotary positions, or causal windowed
attention 'mhsa(N)' over a K / V ring, dim_hidden 96,
dim_ffn 192, dim_squeeze 8, 4 heads, kernel sizes (5, 3), conv groups (8, 8), encoder kernel 5, norms LN/LN/GN/LN/LN/LN; everything else
raises NotImplementedError (callers fall back to OnlineStreamer, the torch.nn step)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import Tensor

from . import ops
from ._lib import NBSS_F32, Lib, hip, make_cfg


def supported(net) -> Optional[str]:
    """None when `net` fits the native step, else the reason"""
    import math
    from models.arch.base.retention import MultiScaleRetention, RetNetRelPos
    l0 = net.layers[0]
    r = l0.mhsa
    if isinstance(r, torch.nn.MultiheadAttention):
        if net.rope is not False or getattr(net, "attn_scope", math.inf) is math.inf or net.attn_scope > 4096:
            return "windowed attention needs a finite window 'mhsa(N)' (N <= 4096) and no ALiBi / rotary positions"
        if (r.embed_dim, r.num_heads) != (96, 4) or r.in_proj_bias is None:
            return "attention geometry must be embed 96, 4 heads, with biases"
    else:
        if not isinstance(getattr(net, "pos", None), RetNetRelPos) or net.rope is not False:
            return "attention must be 'mhsa(N)' or retention without rotary positions ('ret(2)' with rope: false)"
        if not isinstance(r, MultiScaleRetention) or (r.embed_dim, r.value_dim, r.num_heads, r.look_ahead) != (96, 192, 4, 0):
            return "retention geometry must be embed 96, value 192 (factor 2), 4 heads, no look-ahead"
    if net.encoder.kernel_size[0] != 5 or net.encoder.in_channels > 32 or l0.squeeze[0].out_channels != 8 or l0.tconvffn[1].out_channels != 192:
        return "geometry must be encoder kernel 5, dim_squeeze 8, dim_ffn 192"
    if l0.tconvffn[3].kernel_size[0] != 3 or l0.tconvffn[3].groups != 8 or l0.fconv1[1].kernel_size[0] != 5 or l0.fconv1[1].groups != 8:
        return "kernel sizes (5, 3) and conv groups (8, 8)"
    kinds = [type(m).__name__ for m in (l0.norm_mhsa, l0.tconvffn[0], l0.tconvffn[6], l0.fconv1[0], l0.fconv2[0], l0.norm_full)]
    if kinds != ["LayerNorm", "LayerNorm", "GroupNorm", "LayerNorm", "LayerNorm", "LayerNorm"]:
        return f"norms must be LN, LN, GN, LN, LN, LN (got {kinds})"
    if any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in net.modules()):
        return "dropout must be 0"
    return None


class NativeOnlineStreamer:
    def __init__(self, net, batch: int, chunk: int, device=None, use_graph: Optional[bool] = None, lib: Optional[Lib] = None):
        why = supported(net)
        if why is not None:
            raise NotImplementedError("native OnlineSpatialNet step: " + why)
        if not 0 < chunk <= 32:
            raise NotImplementedError("native OnlineSpatialNet step: 1..32 frames per chunk")
        self.net, self.B, self.C = net.eval(), batch, chunk
        self.dev = torch.device(device) if device is not None else net.decoder.weight.device
        self.lib = lib if lib is not None else hip()  # (tests pass the host emulator build)
        self.use_graph = self.dev.type == "cuda" if use_graph is None else use_graph
        self.F, self.din, self.dout, self.L = net.layers[0].full.in_features, net.encoder.in_channels, net.decoder.out_features, len(net.layers)
        sd = {k: v.detach().float() for k, v in net.state_dict().items()}
        # the cross-band kernels read SpatialNet's flat parameter buffer / packed fragments: same names for everything they touch;
        # the attention slots of that layout stay zero (the retention weights go to the native kernel directly)
        # (layers 0..full_share own a LinearGroup, the later ones share the last of them: OnlineSpatialNet.__init__)
        self.cfg = make_cfg(batch, self.F, chunk, self.din, self.dout, L=self.L, dtype=NBSS_F32, full_share=len({id(l.full) for l in net.layers}) - 1)
        p: Dict[str, Tensor] = {}
        from .params import param_specs
        for name, shape in param_specs(self.cfg):
            p[name] = sd[name] if (name in sd and ".mhsa." not in name) else torch.zeros(shape)
        self.flat = ops.flatten_params(self.lib, self.cfg, p, self.dev)
        self.packed = ops.pack_params(self.lib, self.cfg, self.flat)
        dv = lambda t: t.contiguous().to(self.dev)  # noqa: E731
        self.enc_w, self.enc_b = dv(sd["encoder.weight"]), dv(sd["encoder.bias"])
        self.windowed = isinstance(net.layers[0].mhsa, torch.nn.MultiheadAttention)
        self.scope = int(net.attn_scope) if self.windowed else 0
        self.ring = self.scope - 1 + chunk
        self.decay = None if self.windowed else dv(net.pos.decay.detach().float().exp())
        self.layers = []
        for l in range(self.L):
            q = f"layers.{l}."
            tc = q + "tconvffn."
            att = ({"win_t": dv(sd[q + "mhsa.in_proj_weight"].t()), "bin": dv(sd[q + "mhsa.in_proj_bias"]), "wo_t": dv(sd[q + "mhsa.out_proj.weight"].t()),
                    "bo": dv(sd[q + "mhsa.out_proj.bias"])} if self.windowed else
                   {"wq_t": dv(sd[q + "mhsa.q_proj.weight"].t()), "wk_t": dv(sd[q + "mhsa.k_proj.weight"].t()) if (q + "mhsa.k_proj.weight") in sd else None,
                    "wv_t": dv(sd[q + "mhsa.v_proj.weight"].t()), "wg_t": dv(sd[q + "mhsa.g_proj.weight"].t()), "wo_t": dv(sd[q + "mhsa.out_proj.weight"].t())})
            self.layers.append({
                **att,
                "ln": (dv(sd[q + "norm_mhsa.weight"]), dv(sd[q + "norm_mhsa.bias"])),
                "tln": (dv(sd[tc + "0.weight"]), dv(sd[tc + "0.bias"])), "w1_t": dv(sd[tc + "1.weight"][:, :, 0].t()), "b1": dv(sd[tc + "1.bias"]),
                "c1": (dv(sd[tc + "3.weight"]), dv(sd[tc + "3.bias"])), "c2": (dv(sd[tc + "5.weight"]), dv(sd[tc + "5.bias"])),
                "gn": (dv(sd[tc + "6.weight"]), dv(sd[tc + "6.bias"])), "c3": (dv(sd[tc + "8.weight"]), dv(sd[tc + "8.bias"])),
                "w2_t": dv(sd[tc + "10.weight"][:, :, 0].t()), "b2": dv(sd[tc + "10.bias"]),
            })
        BF, z = batch * self.F, lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.dev)  # noqa: E731
        self.x, self.y = z(batch, self.F, chunk, self.din), z(batch, self.F, chunk, self.dout)
        self.h = [z(batch, self.F, chunk, 96), z(batch, self.F, chunk, 96)]
        self.a3, self.gn_sums = z(BF, chunk, 192), z(batch + BF, chunk, 8, 2)  # GroupNorm sums [B] + per-frequency partials [B*F] (fixed-order fold)
        self.state = {"enc": z(BF, 4, self.din), "s": [[z(BF, 2, 192) for _ in range(3)] for _ in range(self.L)]}
        if self.windowed:  # K / V rings of the last scope - 1 + chunk frames per layer, one device-side frame counter for the stream
            self.state["kring"] = [z(BF, self.ring, 96) for _ in range(self.L)]
            self.state["vring"] = [z(BF, self.ring, 96) for _ in range(self.L)]
            self.state["pos"] = torch.zeros(1, dtype=torch.int32, device=self.dev)
        else:
            self.state["kv"] = [z(BF, 4, 24, 48) for _ in range(self.L)]
            self.state["scale"] = [z(BF, 4) for _ in range(self.L)]
        self.graph = None

    def _buffers(self):
        out = [self.state["enc"]]
        out += (self.state["kring"] + self.state["vring"] + [self.state["pos"]]) if self.windowed else (self.state["kv"] + self.state["scale"])
        for s in self.state["s"]:
            out += s
        return out

    @torch.no_grad()
    def reset(self) -> None:
        for b in self._buffers():
            b.zero_()

    def _run(self) -> None:
        """one step: reads self.x, writes self.y, updates every state buffer in place (a fixed sequence of 6 L + 2 C-ABI calls)"""
        lib, cfg, P = self.lib, C.byref(self.cfg), ops._ptr
        st = ops._stream(lib, self.x)
        BF, Cc = self.B * self.F, self.C
        f = lambda t: P(lib, t, torch.float32)  # noqa: E731
        a, b = self.h
        lib.call("nbss_online_encoder_step", BF, Cc, self.din, f(self.enc_w), f(self.enc_b), f(self.x), f(self.state["enc"]), f(a), st)
        for l, w in enumerate(self.layers):
            lib.call("nbss_fconv_fwd", cfg, f(self.flat), P(lib, self.packed), l, 0, f(a), f(b), st)
            lib.call("nbss_full_fwd", cfg, f(self.flat), P(lib, self.packed), l, f(b), f(a), st)
            lib.call("nbss_fconv_fwd", cfg, f(self.flat), P(lib, self.packed), l, 1, f(a), f(b), st)
            if self.windowed:
                lib.call("nbss_online_mhsa_step", BF, Cc, self.scope, self.ring, f(w["ln"][0]), f(w["ln"][1]), f(w["win_t"]), f(w["bin"]), f(w["wo_t"]), f(w["bo"]),
                         f(self.state["kring"][l]), f(self.state["vring"][l]), P(lib, self.state["pos"], torch.int32), f(b), st)
            else:
                lib.call("nbss_online_ret_step", BF, Cc, f(w["ln"][0]), f(w["ln"][1]), f(w["wq_t"]), f(w["wk_t"]) if w["wk_t"] is not None else None, f(w["wv_t"]),
                         f(w["wg_t"]), f(w["wo_t"]), f(self.decay), f(self.state["kv"][l]), f(self.state["scale"][l]), f(b), st)
            s1, s2, s3 = self.state["s"][l]
            lib.call("nbss_online_tconvffn_step", self.B, self.F, Cc, f(w["tln"][0]), f(w["tln"][1]), f(w["w1_t"]), f(w["b1"]), f(w["c1"][0]), f(w["c1"][1]),
                     f(w["c2"][0]), f(w["c2"][1]), f(w["gn"][0]), f(w["gn"][1]), f(w["c3"][0]), f(w["c3"][1]), f(w["w2_t"]), f(w["b2"]), f(s1), f(s2), f(s3),
                     f(self.a3), f(self.gn_sums), f(b), st)
            a, b = b, a
        if self.windowed:
            lib.call("nbss_online_advance", P(lib, self.state["pos"], torch.int32), Cc, st)
        lib.call("nbss_decoder_fwd", cfg, f(self.flat), P(lib, self.packed), f(a), f(self.y), st)

    def _capture(self) -> None:
        saved = [b.clone() for b in self._buffers()]
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up outside the capture: code-object loads, LDS attribute calls
                self._run()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()  # a HIP graph on ROCm
        with torch.cuda.graph(self.graph):
            self._run()
        for b, s0 in zip(self._buffers(), saved):  # warm-up and capture advanced the state: rewind
            b.copy_(s0)

    @torch.no_grad()
    def step(self, x_chunk: Tensor) -> Tensor:
        self.x.copy_(x_chunk)
        if self.use_graph:
            if self.graph is None:
                self._capture()
            self.graph.replay()
        else:
            self._run()
        return self.y.clone()
