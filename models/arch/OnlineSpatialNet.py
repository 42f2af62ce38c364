"""OnlineSpatialNet — drop-in for the reference's models/arch/OnlineSpatialNet.py (constructor :261-282, forward(x, inference=False,
return_attn_score=False) :333, same state_dict keys): the causal variant of SpatialNet for streaming speech separation.
Differences to SpatialNet: the encoder and the three T-convs of the T-ConvFFN are causal (left padding, or the last k-1 input
frames from a state), the GroupNorm inside the T-ConvFFN normalises each FRAME over (channels of a group x frequencies) instead of
a sequence over time, and the narrow-band attention is causal:
    'mhsa(N)'  masked multi-head self-attention over the last N frames ('inf': all past frames; rope='ALiBi' adds linear biases),
    'ret(2)'   multi-scale retention (models/arch/base/retention.py), parallel / chunkwise for training, recurrent for streaming,
    'mamba(..)' needs mamba_ssm, which is neither pinned by the reference nor installed here: raises.
Besides the reference's whole-utterance forward this module offers an explicit streaming interface — `init_stream(batch)` /
`forward_stream(x_chunk, state)` — that carries conv states, a K/V ring (mhsa) or the retention state across chunks; with fixed
chunk shapes a step is a static kernel sequence and can be captured in a HIP graph (`OnlineStreamer`).
Plain PyTorch (BASELINE config 5)."""
import math
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor
from torch.nn import MultiheadAttention

from models.arch.base.linear_group import LinearGroup
from models.arch.base.norm import new_norm
from models.arch.base.retention import MultiScaleRetention, RetNetRelPos


class CausalConv1d(nn.Conv1d):
    """Conv1d over [B,C,T] that sees `look_ahead` future frames at most; `state` (dict keyed by id(self)) carries the last k-1 frames"""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1, groups: int = 1, bias: bool = True,
                 padding_mode: str = "zeros", device=None, dtype=None, look_ahead: int = 0) -> None:
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode, device, dtype)
        self.look_ahead = look_ahead
        assert look_ahead <= self.kernel_size[0] - 1, (look_ahead, self.kernel_size)

    def forward(self, x: Tensor, state: Dict[int, Any] = None) -> Tensor:
        k = self.kernel_size[0]
        if state is None or id(self) not in state:
            x = F.pad(x, (k - 1 - self.look_ahead, self.look_ahead))
        else:
            x = torch.cat([state[id(self)], x], dim=-1)
        if state is not None:
            state[id(self)] = x[..., x.shape[-1] - (k - 1):]
        return super().forward(x)

    def extra_repr(self):
        return super().extra_repr() + (f", look ahead={self.look_ahead}" if self.look_ahead else "")


class SpatialNetLayer(nn.Module):
    def __init__(self, dim_hidden: int, dim_ffn: int, dim_squeeze: int, num_freqs: int, num_heads: int, dropout: Tuple[float, float, float] = (0, 0, 0),
                 kernel_size: Tuple[int, int] = (5, 3), conv_groups: Tuple[int, int] = (8, 8), norms: List[str] = ("LN", "LN", "GN", "LN", "LN", "LN"),
                 padding: str = "zeros", full: nn.Module = None, attention: str = "mhsa") -> None:
        super().__init__()
        fg, tg = conv_groups
        fk, tk = kernel_size

        def fconv(norm):
            return nn.ModuleList([new_norm(norm, dim_hidden, seq_last=True, group_size=None, num_groups=fg),
                                  nn.Conv1d(dim_hidden, dim_hidden, kernel_size=fk, groups=fg, padding="same", padding_mode=padding), nn.PReLU(dim_hidden)])

        self.fconv1 = fconv(norms[3])
        self.norm_full = new_norm(norms[5], dim_hidden, seq_last=False, group_size=None, num_groups=fg)
        self.full_share = full is not None
        self.squeeze = nn.Sequential(nn.Conv1d(dim_hidden, dim_squeeze, kernel_size=1), nn.SiLU())
        self.dropout_full = nn.Dropout2d(dropout[2]) if dropout[2] > 0 else None
        self.full = LinearGroup(num_freqs, num_freqs, num_groups=dim_squeeze) if full is None else full
        self.unsqueeze = nn.Sequential(nn.Conv1d(dim_squeeze, dim_hidden, kernel_size=1), nn.SiLU())
        self.fconv2 = fconv(norms[4])
        self.norm_mhsa = new_norm(norms[0], dim_hidden, seq_last=False, group_size=None, num_groups=tg)
        if attention.startswith("ret"):  # ret(<value factor>,share_qk|not_share_qk)
            factor, share = attention[4:-1].split(",")
            assert share in ("share_qk", "not_share_qk"), attention
            self.mhsa = MultiScaleRetention(embed_dim=dim_hidden, num_heads=num_heads, value_factor=int(factor), share_qk=share == "share_qk")
        elif attention.startswith("mamba"):
            raise NotImplementedError("attention='mamba(..)' needs mamba_ssm (unpinned in the reference's requirements.txt, not installed): use mhsa(N) or ret(2)")
        else:
            self.mhsa = MultiheadAttention(embed_dim=dim_hidden, num_heads=num_heads, batch_first=True)
        self.attention = attention
        self.dropout_mhsa = nn.Dropout(dropout[0])

        def tconv():
            return CausalConv1d(dim_ffn, dim_ffn, kernel_size=tk, groups=tg)

        self.tconvffn = nn.ModuleList([new_norm(norms[1], dim_hidden, seq_last=True, group_size=None, num_groups=tg), nn.Conv1d(dim_hidden, dim_ffn, kernel_size=1),
                                       nn.SiLU(), tconv(), nn.SiLU(), tconv(), new_norm(norms[2], dim_ffn, seq_last=True, group_size=None, num_groups=tg),
                                       nn.SiLU(), tconv(), nn.SiLU(), nn.Conv1d(dim_ffn, dim_hidden, kernel_size=1)])
        self.dropout_tconvffn = nn.Dropout(dropout[1])

    def forward(self, x: Tensor, att_mask=None, chunkwise_recurrent: bool = True, rope=True, state: Dict[int, Any] = None, inference: bool = False):
        """x [B,F,T,H] -> (x, attention weights or None)"""
        x = x + self._fconv(self.fconv1, x)
        x = x + self._full(x)
        x = x + self._fconv(self.fconv2, x)
        a, attn = self._tsa(x, att_mask, chunkwise_recurrent, rope, state=state, inference=inference)
        x = x + a
        x = x + self._tconvffn(x, state=state)
        return x, attn

    # ---- narrow-band -----------------------------------------------------------------------------------------------------------
    def _tsa(self, x: Tensor, attn_mask, chunkwise_recurrent: bool, rope=True, state: Dict[int, Any] = None, inference: bool = False):
        B, Fq, T, H = x.shape
        u = self.norm_mhsa(x).reshape(B * Fq, T, H)
        attn = None
        if isinstance(self.mhsa, MultiheadAttention):
            u, attn = self.mhsa(u, u, u, need_weights=bool(getattr(self, "need_weights", False)), average_attn_weights=False, attn_mask=attn_mask)
        elif not inference:
            u = self.mhsa(u, rel_pos=attn_mask, incremental_state=state, chunkwise_recurrent=chunkwise_recurrent, rope=rope)
        else:  # frame-by-frame recurrence (what a streaming deployment computes)
            st: Dict[str, Any] = {}
            u = torch.cat([self.mhsa(u[:, i:i + 1], rel_pos=attn_mask[i], incremental_state=st, rope=rope) for i in range(T)], dim=1)
        return self.dropout_mhsa(u.reshape(B, Fq, T, H)), attn

    def _tconvffn(self, x: Tensor, state: Dict[int, Any] = None) -> Tensor:
        B, Fq, T, H0 = x.shape
        h = x.transpose(-1, -2).reshape(B * Fq, H0, T)
        for m in self.tconvffn:
            if isinstance(m, CausalConv1d):
                h = m(h, state=state)
            elif "GroupNorm" in type(m).__name__:  # per frame, over (channels of a group) x (all frequencies)
                C = h.shape[1]
                h = m(h.reshape(B, Fq, C, T).transpose(1, -1).reshape(B * T, C, Fq))
                h = h.reshape(B, T, C, Fq).transpose(1, -1).reshape(B * Fq, C, T)
            else:
                h = m(h)
        return self.dropout_tconvffn(h.reshape(B, Fq, H0, T).transpose(-1, -2))

    # ---- cross-band ------------------------------------------------------------------------------------------------------------
    def _fconv(self, ml: nn.ModuleList, x: Tensor) -> Tensor:
        B, Fq, T, H = x.shape
        h = x.permute(0, 2, 3, 1).reshape(B * T, H, Fq)
        for m in ml:
            h = m(h)
        return h.reshape(B, T, H, Fq).permute(0, 3, 1, 2)

    def _full(self, x: Tensor) -> Tensor:
        B, Fq, T, H = x.shape
        h = self.squeeze(self.norm_full(x).permute(0, 2, 3, 1).reshape(B * T, H, Fq))
        if self.dropout_full:
            h = self.dropout_full(h.reshape(B, T, -1, Fq).transpose(1, 3)).transpose(1, 3).reshape(B * T, -1, Fq)
        h = self.unsqueeze(self.full(h))
        return h.reshape(B, T, H, Fq).permute(0, 3, 1, 2)

    def extra_repr(self) -> str:
        return f"full_share={self.full_share}"


class OnlineSpatialNet(nn.Module):
    def __init__(self, dim_input: int, dim_output: int, num_layers: int, dim_squeeze: int, num_freqs: int, encoder_kernel_size: int = 5,
                 dim_hidden: int = 192, dim_ffn: int = 384, num_heads: int = 2, dropout: Tuple[float, float, float] = (0, 0, 0),
                 kernel_size: Tuple[int, int] = (5, 3), conv_groups: Tuple[int, int] = (8, 8), norms: List[str] = ("LN", "LN", "GN", "LN", "LN", "LN"),
                 padding: str = "zeros", full_share: int = 0, attention: str = "mhsa(251)", decay: Union[int, bool, List[int], List[float]] = 5,
                 chunkwise_recurrent: bool = True, rope: Union[bool, str] = False):
        super().__init__()
        assert attention.startswith(("mhsa", "ret", "mamba")), attention
        assert rope in (True, False, "ALiBi"), rope
        if attention == "ret(2)":  # older checkpoints: Q and K are shared exactly when no rotary encoding is used
            attention = "ret(2,share_qk)" if rope is False else "ret(2,not_share_qk)"
        self.num_heads, self.chunkwise_recurrent, self.rope = num_heads, chunkwise_recurrent, rope
        self.pos = None
        if attention.startswith("ret"):
            self.pos = RetNetRelPos(embed_dim=dim_hidden, num_heads=num_heads, recurrent_chunk_size=64, decay=decay)
        elif attention.startswith("mhsa"):
            self.attn_scope = math.inf if attention[5:-1] == "inf" else int(attention[5:-1])
        self.encoder = CausalConv1d(dim_input, dim_hidden, kernel_size=encoder_kernel_size, look_ahead=0)
        full, layers = None, []
        for l in range(num_layers):
            layer = SpatialNetLayer(dim_hidden=dim_hidden, dim_ffn=dim_ffn, dim_squeeze=dim_squeeze, num_freqs=num_freqs, num_heads=num_heads,
                                    dropout=dropout, kernel_size=kernel_size, conv_groups=conv_groups, norms=norms, padding=padding,
                                    full=full if l > full_share else None, attention=attention)
            full = layer.full
            layers.append(layer)
        self.layers = nn.ModuleList(layers)
        self.decoder = nn.Linear(dim_hidden, dim_output)

    def forward(self, x: Tensor, inference: bool = False, return_attn_score: bool = False):
        """x [B,F,T,dim_input] -> [B,F,T,dim_output]; inference=True evaluates retention frame by frame (recurrent form)"""
        B, Fq, T, H0 = x.shape
        h = self.encoder(x.reshape(B * Fq, T, H0).transpose(1, 2)).transpose(1, 2).reshape(B, Fq, T, -1)
        chunkwise = True if not inference else self.chunkwise_recurrent
        mask = self.get_causal_mask(slen=T, device=x.device, chunkwise_recurrent=chunkwise, batch_size=B * Fq, inference=inference)
        attns = [] if return_attn_score else None
        for layer in self.layers:
            layer.need_weights = return_attn_score
            h, attn = layer(h, mask, chunkwise, self.rope, None, inference)
            if return_attn_score:
                attns.append(attn)
        y = self.decoder(h).contiguous()
        return (y, attns) if return_attn_score else y

    def get_causal_mask(self, slen: int, device=None, chunkwise_recurrent: bool = True, batch_size: int = None, inference: bool = False, offset: int = 0,
                        klen: int = None):
        """mhsa: additive [slen, klen] mask (0 inside the causal window of `attn_scope` frames, -inf outside; query i is frame
        offset + i, key j is frame offset + slen - klen + j); retention: the RetNetRelPos package for the chosen evaluation order"""
        if isinstance(self.pos, RetNetRelPos):
            if not inference:
                return self.pos(slen=slen, chunkwise_recurrent=chunkwise_recurrent)
            return [self.pos(slen=t, activate_recurrent=True) for t in range(slen)]
        klen = slen if klen is None else klen
        qi = torch.arange(slen, device=device)[:, None] + (klen - slen)
        rel = qi - torch.arange(klen, device=device)[None, :]  # how many frames the key lies in the past
        inside = (rel >= 0) & (rel < self.attn_scope)
        if self.rope == "ALiBi":
            assert batch_size is not None
            slopes = (2.0 ** (-8 / torch.arange(1, self.num_heads + 1, device=device))).reshape(self.num_heads, 1, 1)
            bias = torch.where(inside, -rel.abs().float(), torch.full((), -torch.inf, device=device))
            return (slopes * bias).repeat(batch_size, 1, 1)  # [batch * heads, slen, klen]
        return torch.where(inside, 0.0, -torch.inf)

    # ---- streaming -------------------------------------------------------------------------------------------------------------
    def init_stream(self, batch: int, device=None, dtype=torch.float32) -> Dict[str, Any]:
        """empty state for `forward_stream`: causal-conv tails, per-layer K/V ring (mhsa: the last attn_scope - 1 frames, with a
        device-side count of the slots that hold a frame) or retention state, frame counter.  Every tensor has a fixed shape (except
        the K/V cache of 'mhsa(inf)', which grows), so a step can be replayed from a HIP graph."""
        dev = device if device is not None else self.decoder.weight.device
        Fq, H, n = self.layers[0].full.in_features, self.decoder.in_features, batch * self.layers[0].full.in_features
        st: Dict[str, Any] = {"t": 0, "conv": {}, "attn": []}
        for m in self.modules():
            if isinstance(m, CausalConv1d):
                st["conv"][id(m)] = torch.zeros(n, m.in_channels, m.kernel_size[0] - 1, device=dev, dtype=dtype)
        for layer in self.layers:
            if isinstance(layer.mhsa, MultiheadAttention):
                keep = 0 if self.attn_scope is math.inf else self.attn_scope - 1
                st["attn"].append({"k": torch.zeros(n, keep, H, device=dev, dtype=dtype), "v": torch.zeros(n, keep, H, device=dev, dtype=dtype),
                                   "valid": torch.zeros(1, dtype=torch.long, device=dev)})
            else:  # zero state with zero running scale == "no frame seen yet" in MultiScaleRetention.recurrent_forward
                r = layer.mhsa
                st["attn"].append({"prev_key_value": torch.zeros(n, r.num_heads, r.key_dim, r.head_dim, device=dev, dtype=dtype),
                                   "scale": torch.zeros(r.num_heads, device=dev, dtype=dtype)})
        return st

    def forward_stream(self, x: Tensor, state: Dict[str, Any]) -> Tensor:
        """x [B,F,C,dim_input]: the next C frames; equals the corresponding frames of forward() on the whole signal"""
        B, Fq, C, H0 = x.shape
        conv = state["conv"]
        h = self.encoder(x.reshape(B * Fq, C, H0).transpose(1, 2), state=conv).transpose(1, 2).reshape(B, Fq, C, -1)
        t0 = state["t"]
        for layer, ast in zip(self.layers, state["attn"]):
            h = h + layer._fconv(layer.fconv1, h)
            h = h + layer._full(h)
            h = h + layer._fconv(layer.fconv2, h)
            u = layer.norm_mhsa(h).reshape(B * Fq, C, -1)
            if isinstance(layer.mhsa, MultiheadAttention):
                u = self._mhsa_stream(layer.mhsa, u, ast)
            else:
                u = torch.cat([layer.mhsa(u[:, i:i + 1], rel_pos=self.pos(slen=t0 + i + 1, activate_recurrent=True), incremental_state=ast, rope=self.rope)
                               for i in range(C)], dim=1)
            h = h + u.reshape(B, Fq, C, -1)
            h = h + layer._tconvffn(h, state=conv)
        state["t"] = t0 + C
        return self.decoder(h).contiguous()

    def _mhsa_stream(self, mha: MultiheadAttention, u: Tensor, ast: Dict[str, Any]) -> Tensor:
        """causal windowed attention of C new frames against the cached keys / values of the last attn_scope - 1 frames"""
        n, C, H = u.shape
        nh, dh = mha.num_heads, H // mha.num_heads
        q, k, v = F.linear(u, mha.in_proj_weight, mha.in_proj_bias).chunk(3, dim=-1)
        K, V = torch.cat([ast["k"], k], dim=1), torch.cat([ast["v"], v], dim=1)
        L = K.shape[1]
        mask = self.get_causal_mask(slen=C, device=u.device, batch_size=n, klen=L)
        # ring slots that have not been filled yet hold no frame (device-side count: the step stays a static kernel sequence)
        empty = (L - C) - ast["valid"]
        mask = mask + torch.where(torch.arange(L, device=u.device) < empty, -torch.inf, 0.0)
        qh, kh, vh = (z.reshape(n, -1, nh, dh).transpose(1, 2) for z in (q, K, V))
        am = mask if mask.dim() == 2 else mask.reshape(n, nh, C, L)
        o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=am).transpose(1, 2).reshape(n, C, H)
        if self.attn_scope is math.inf:
            ast["k"], ast["v"], ast["valid"] = K, V, ast["valid"] + C
        else:
            keep = self.attn_scope - 1
            ast["k"], ast["v"], ast["valid"] = K[:, L - keep:], V[:, L - keep:], torch.clamp(ast["valid"] + C, max=keep)
        return F.linear(o, mha.out_proj.weight, mha.out_proj.bias)


class OnlineStreamer:
    """Fixed-shape streaming step of an OnlineSpatialNet ('mhsa(N)' with finite N, or 'ret(..)' without rotary positions): every call
    consumes `chunk` frames.  The whole state lives in pre-allocated device buffers that a step updates in place, so a step is a
    static kernel sequence; on a HIP device it is captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed — BASELINE
    config 5's "causal chunked inference with hipGraph-captured steps"."""

    def __init__(self, net: OnlineSpatialNet, batch: int, chunk: int, device=None, use_graph: Optional[bool] = None):
        self.net, self.B, self.C = net.eval(), batch, chunk
        self.dev = torch.device(device) if device is not None else net.decoder.weight.device
        if getattr(net, "attn_scope", 0) is math.inf or (net.pos is not None and net.rope is not False):
            raise NotImplementedError("OnlineStreamer needs a fixed-size state: mhsa(N) with finite N, or retention without rotary positions")
        self.Fq, self.din = net.layers[0].full.in_features, net.encoder.in_channels
        self.use_graph = self.dev.type == "cuda" if use_graph is None else use_graph
        self.x = torch.zeros(batch, self.Fq, chunk, self.din, device=self.dev)
        self.y = torch.zeros(batch, self.Fq, chunk, net.decoder.out_features, device=self.dev)
        self.state = net.init_stream(batch, device=self.dev)
        self.graph = None

    def _buffers(self):
        out = list(self.state["conv"].values())
        for a in self.state["attn"]:
            out += list(a.values())
        return out

    @torch.no_grad()
    def _run(self) -> None:
        """one step: reads self.x, writes self.y, updates every state buffer in place"""
        st = {"t": 0, "conv": dict(self.state["conv"]), "attn": [dict(a) for a in self.state["attn"]]}
        y = self.net.forward_stream(self.x, st)
        for key, buf in self.state["conv"].items():
            buf.copy_(st["conv"][key])
        for new, buf in zip(st["attn"], self.state["attn"]):
            for k2 in buf:
                buf[k2].copy_(new[k2])
        self.y.copy_(y)

    def _capture(self) -> None:
        saved = [b.clone() for b in self._buffers()]
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._run()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._run()
        for b, s0 in zip(self._buffers(), saved):  # warm-up and capture advanced the state: rewind
            b.copy_(s0)

    @torch.no_grad()
    def reset(self) -> None:
        """back to the empty state (a new utterance); the captured graph stays valid: it only refers to the buffers"""
        for b in self._buffers():
            b.zero_()

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
