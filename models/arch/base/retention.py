"""Multi-scale retention for the online SpatialNet — drop-in for the reference's models/arch/base/retention.py (RetNetRelPos
:36-99, MultiScaleRetention :128-300; same constructor arguments, parameter names and forward signature).

Retention of one head with decay g:  o_t = sum_{s<=t} g^(t-s) (q_t . k_s) v_s, followed by a per-head RMS normalisation without
affine and a SiLU gate.  Because of that normalisation any positive per-(head, frame) scaling of o_t is immaterial, which is
what lets the same function be evaluated in three ways:
  * parallel   — one masked [T,T] product per head (training; the decay mask rows are divided by the square root of their sums
                 and the score rows by their clamped absolute sums, as in the reference, for fp16/bf16 range),
  * recurrent  — a [dk,dv] state per head, S_t = g S_{t-1} + k_t^T v_t, o_t = q_t S_t (frame-by-frame streaming),
  * chunkwise  — parallel inside chunks of `recurrent_chunk_size` frames, the recurrent state across chunks.
Plain PyTorch."""
from typing import Any, Dict, Iterable, List, Union

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6, elementwise_affine=True):
        super().__init__()
        self.eps, self.elementwise_affine = eps, elementwise_affine
        if elementwise_affine:
            self.weight = nn.Parameter(torch.ones(dim))
        else:
            self.register_parameter("weight", None)

    def forward(self, x):
        y = (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x)
        return y if self.weight is None else y * self.weight


class RetNetRelPos(nn.Module):
    """positions for retention: rotary angles and the per-head decay, packaged for the three evaluation modes"""

    def __init__(self, embed_dim: int, num_heads: int, recurrent_chunk_size: int, decay: Union[int, bool, List[int], List[float]] = None):
        super().__init__()
        half = embed_dim // num_heads // 2
        angle = (1.0 / (10000 ** torch.linspace(0, 1, half))).repeat_interleave(2)
        if decay is False:
            self.decays = [1] * num_heads
        elif isinstance(decay, Iterable):
            if isinstance(decay[0], float):
                assert decay[0] <= 1, decay
                self.decays = list(decay)
            else:
                assert isinstance(decay[0], int) and decay[0] > 1, decay
                self.decays = [1 - 2.0 ** (-d) for d in decay]
        else:
            if decay is None or decay is True:
                decay = 5
            self.decays = (1 - 2 ** (-decay - torch.arange(num_heads, dtype=torch.float))).tolist()
        self.register_buffer("angle", angle)
        self.register_buffer("decay", torch.log(torch.tensor(self.decays, dtype=torch.float)))
        self.recurrent_chunk_size = recurrent_chunk_size

    def _rot(self, index: Tensor):
        ang = index[:, None] * self.angle[None, :]
        return torch.sin(ang), torch.cos(ang)

    def _decay_mask(self, n: int) -> Tensor:
        """[heads, n, n]: g^(i-j) for j <= i, 0 above the diagonal"""
        i = torch.arange(n, device=self.decay.device, dtype=self.decay.dtype)
        d = i[:, None] - i[None, :]
        return torch.where(d >= 0, torch.exp(d.clamp(min=0) * self.decay[:, None, None]), torch.zeros((), device=d.device, dtype=self.decay.dtype))

    def forward(self, slen: int, activate_recurrent: bool = False, chunkwise_recurrent: bool = False):
        if activate_recurrent:  # frame index slen - 1 (the reference's convention: called with slen = t gives angle * (t - 1))
            pos = self.angle * (slen - 1)
            return (torch.sin(pos), torch.cos(pos)), self.decay.exp()
        index = torch.arange(slen, device=self.decay.device, dtype=self.decay.dtype)
        if chunkwise_recurrent:
            return self._rot(index), ("chunk", self.recurrent_chunk_size, self.decay.exp())
        mask = self._decay_mask(slen)
        return self._rot(index), mask / mask.sum(dim=-1, keepdim=True).sqrt()

    def extra_repr(self) -> str:
        return f"decays={self.decays} -> effective len={[1 / (1 - d) if d < 1 else float('inf') for d in self.decays]}"

    def _load_from_state_dict(self, *args, **kwargs):  # constants (the reference skips loading them as well)
        return


def rotate_every_two(x: Tensor) -> Tensor:
    return torch.stack((-x[..., 1::2], x[..., ::2]), dim=-1).flatten(-2)


def theta_shift(x: Tensor, sin: Tensor, cos: Tensor) -> Tensor:
    n = x.shape[-2]
    return x * cos[:n] + rotate_every_two(x) * sin[:n]


class MultiScaleRetention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, value_factor: int = 2, gate_fn: str = "swish", look_ahead: int = 0, share_qk: bool = False):
        super().__init__()
        self.embed_dim, self.value_dim, self.num_heads = embed_dim, embed_dim * value_factor, num_heads
        self.head_dim, self.key_dim = self.value_dim // num_heads, embed_dim // num_heads
        self.scaling = self.key_dim ** -0.5
        self.look_ahead, self.share_qk = look_ahead, share_qk
        if gate_fn not in ("swish", "gelu"):
            raise NotImplementedError(gate_fn)
        self.gate_fn = F.silu if gate_fn == "swish" else F.gelu
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        self.k_proj = None if share_qk else nn.Linear(embed_dim, embed_dim, bias=False)
        self.v_proj = nn.Linear(embed_dim, self.value_dim, bias=False)
        self.g_proj = nn.Linear(embed_dim, self.value_dim, bias=False)
        self.out_proj = nn.Linear(self.value_dim, embed_dim, bias=False)
        self.group_norm = RMSNorm(self.head_dim, eps=1e-6, elementwise_affine=False)
        self.reset_parameters()

    def reset_parameters(self):
        for lin, gain in ((self.q_proj, 2 ** -2.5), (self.k_proj, 2 ** -2.5), (self.v_proj, 2 ** -2.5), (self.g_proj, 2 ** -2.5), (self.out_proj, 2 ** -1)):
            if lin is not None:
                nn.init.xavier_uniform_(lin.weight, gain=gain)

    # ---- the three evaluation orders (q, k: [B,heads,T,dk]; v: [B,T,value_dim]) -> [B,T,heads,dv] -------------------------
    def parallel_forward(self, qr: Tensor, kr: Tensor, v: Tensor, mask: Tensor) -> Tensor:
        B, T, _ = v.shape
        vr = v.view(B, T, self.num_heads, self.head_dim).transpose(1, 2)
        s = (qr @ kr.transpose(-1, -2)) * mask
        s = s / s.detach().abs().sum(dim=-1, keepdim=True).clamp(min=1, max=5e4)
        return (s @ vr).transpose(1, 2)

    def recurrent_forward(self, qr: Tensor, kr: Tensor, v: Tensor, decay: Tensor, incremental_state: Dict[str, Any]) -> Tensor:
        """one frame: qr, kr [B,heads,1,dk], v [B,1,value_dim]; state: "prev_key_value" [B,heads,dk,dv] (+ its running scale)"""
        B = v.shape[0]
        kv = kr.transpose(-1, -2) * v.view(B, self.num_heads, 1, self.head_dim)  # k_t^T v_t
        g = decay.view(1, self.num_heads, 1, 1)
        if "prev_key_value" in incremental_state:  # kept normalised by (1 + g + g^2 + ...)^(1/2) like the parallel mask rows
            prev_scale = incremental_state["scale"]
            scale = prev_scale * decay + 1
            carry = (prev_scale.sqrt() * decay / scale.sqrt()).view(1, self.num_heads, 1, 1)
            kv = incremental_state["prev_key_value"] * carry + kv / scale.sqrt().view(1, self.num_heads, 1, 1)
        else:
            scale = torch.ones_like(decay)
        incremental_state["prev_key_value"], incremental_state["scale"] = kv, scale
        return (qr @ kv).transpose(1, 2)  # [B,1,heads,dv]

    def chunk_recurrent_forward(self, qr: Tensor, kr: Tensor, v: Tensor, inner) -> Tensor:
        _, chunk, gamma = inner
        B, T, _ = v.shape
        H, dk, dv = self.num_heads, self.key_dim, self.head_dim
        vr = v.view(B, T, H, dv).transpose(1, 2)  # [B,H,T,dv]
        lg = torch.log(gamma).view(1, H, 1, 1)
        state = torch.zeros(B, H, dk, dv, dtype=v.dtype, device=v.device)
        outs = []
        for c0 in range(0, T, chunk):
            q, k, vv = qr[:, :, c0:c0 + chunk], kr[:, :, c0:c0 + chunk], vr[:, :, c0:c0 + chunk]
            n = q.shape[2]
            i = torch.arange(n, device=v.device, dtype=v.dtype)
            d = i[:, None] - i[None, :]
            mask = torch.where(d >= 0, torch.exp(d.clamp(min=0) * lg), torch.zeros((), device=v.device, dtype=v.dtype))  # [1,H,n,n]
            # the same positive row scalings as the parallel form (decay row sums, clamped absolute score sums), applied to the
            # inner and the cross-chunk part alike: immaterial after the RMS normalisation, but it keeps the magnitudes (hence the
            # effect of its eps) where the reference has them
            rs = mask.sum(dim=-1, keepdim=True).sqrt()
            sc = (q @ k.transpose(-1, -2)) * (mask / rs)
            ab = sc.detach().abs().sum(dim=-1, keepdim=True).clamp(min=1)
            inner_out = (sc / ab) @ vv
            cross = ((q * torch.exp((i + 1).view(1, 1, n, 1) * lg)) @ state) / (rs * ab)
            outs.append(inner_out + cross)
            state = state * torch.exp(n * lg) + (k * torch.exp((n - 1 - i).view(1, 1, n, 1) * lg)).transpose(-1, -2) @ vv
        return torch.cat(outs, dim=2).transpose(1, 2)

    def forward(self, x: Tensor, rel_pos, chunkwise_recurrent: bool = False, incremental_state: Dict[str, Any] = None, rope: bool = True) -> Tensor:
        B, T, _ = x.shape
        (sin, cos), inner = rel_pos
        q = self.q_proj(x).view(B, T, self.num_heads, self.key_dim).transpose(1, 2)
        k = q if self.share_qk else (self.k_proj(x) * self.scaling).view(B, T, self.num_heads, self.key_dim).transpose(1, 2)
        v, g = self.v_proj(x), self.g_proj(x)
        qr, kr = (theta_shift(q, sin, cos), theta_shift(k, sin, cos)) if rope else (q, k)
        if self.look_ahead > 0:
            assert incremental_state is None, "look-ahead is not defined for the recurrent form"
            kr, v_ = F.pad(kr, (0, 0, 0, self.look_ahead)), F.pad(v, (0, 0, 0, self.look_ahead))
            qr = F.pad(qr, (0, 0, self.look_ahead, 0))
        else:
            v_ = v
        if incremental_state is not None:
            out = self.recurrent_forward(qr, kr, v_, inner, incremental_state)
        elif chunkwise_recurrent:
            out = self.chunk_recurrent_forward(qr, kr, v_, inner)
        else:
            out = self.parallel_forward(qr, kr, v_, inner)
        if self.look_ahead > 0:
            out = out[:, :-self.look_ahead]
        out = self.group_norm(out).reshape(B, T, self.head_dim * self.num_heads)
        return self.out_proj(self.gate_fn(g) * out)

    def extra_repr(self) -> str:
        return f"num_heads={self.num_heads}, share_qk={self.share_qk}" + (f", look_ahead={self.look_ahead}" if self.look_ahead > 0 else "")
