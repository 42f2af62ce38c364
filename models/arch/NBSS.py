"""NBSS — drop-in for the reference's models/arch/NBSS.py:20-99: time-domain wrapper
`NBSS(n_channel, n_speaker, n_fft, n_overlap, ref_channel, arch, arch_kwargs).forward(x [B,C,T]) -> [B,Spk,T]` around a
narrow-band network ("NB_BLSTM", "NBC" or "NBC2").  STFT (hann, hop = n_overlap) -> per-frequency normalisation by the mean
magnitude of the reference channel -> network on [B,F,T,2C] -> de-normalisation -> iSTFT.  `neg_si_sdr` is the full-band PIT
criterion's metric (the reference takes SI-SDR from torchmetrics; its closed form is restated here).  On a HIP device the STFT / iSTFT are the
kernels of nbss_amd/csrc/signal.hip (through models.io.stft.STFT) and NBC / NBC2 take their native paths; the CPU runs plain PyTorch."""
from typing import Any, Dict

import torch
import torch.nn as nn
from torch import Tensor

from models.arch.blstm2_fc1 import BLSTM2_FC1
from models.arch.NBC import NBC
from models.arch.NBC2 import NBC2


def si_sdr(preds: Tensor, target: Tensor) -> Tensor:
    """scale-invariant SDR over the last axis (zero_mean=False), in dB"""
    eps = torch.finfo(preds.dtype).eps
    alpha = ((preds * target).sum(-1, keepdim=True) + eps) / ((target * target).sum(-1, keepdim=True) + eps)
    scaled = alpha * target
    return 10 * torch.log10(((scaled ** 2).sum(-1) + eps) / (((scaled - preds) ** 2).sum(-1) + eps))


def neg_si_sdr(preds: Tensor, target: Tensor) -> Tensor:
    """-mean over speakers of SI-SDR -> [batch]"""
    return -si_sdr(preds, target).reshape(target.shape[0], -1).mean(1)


_ARCHS = {"NB_BLSTM": BLSTM2_FC1, "NBC": NBC, "NBC2": NBC2}


class NBSS(nn.Module):
    def __init__(self, n_channel: int = 8, n_speaker: int = 2, n_fft: int = 512, n_overlap: int = 256, ref_channel: int = 0, arch: str = "NB_BLSTM",
                 arch_kwargs: Dict[str, Any] = dict()):
        super().__init__()
        if arch not in _ARCHS:
            raise Exception(f"Unkown arch={arch}")
        self.arch: nn.Module = _ARCHS[arch](dim_input=2 * n_channel, dim_output=2 * n_speaker, **arch_kwargs)
        self.register_buffer("window", torch.hann_window(n_fft), False)
        self.n_fft, self.n_overlap, self.ref_channel, self.n_channel, self.n_speaker = n_fft, n_overlap, ref_channel, n_channel, n_speaker

    def _io(self):
        """models.io.stft.STFT of this geometry — NOT a registered sub-module: the reference's NBSS has no `stft.*` state_dict key (its window is a
        non-persistent buffer), and checkpoints interchange both ways"""
        if "_stft" not in self.__dict__:
            from models.io.stft import STFT
            object.__setattr__(self, "_stft", STFT(self.n_fft, self.n_overlap))
        return self.__dict__["_stft"]

    def forward(self, x: Tensor) -> Tensor:
        B, C, N = x.shape
        io = self._io()
        # HIP device: the DFT-as-GEMM kernels of nbss_amd/csrc/signal.hip (models/io/stft.py) — libnbss_hip.so is MANDATORY there, like under SpatialNet:
        # a missing / unloadable library raises (nbss_amd._lib.hip) instead of silently running torch.stft, so that a GPU run never reports numbers of a
        # path that is not the product's.  Host tensors and STFT geometries outside the kernels' set (hip_ok) take torch.stft / istft.
        native_io = x.is_cuda and io.hip_ok
        if native_io:
            X = io.stft(x.reshape(B * C, N))[0]
        else:
            X = torch.stft(x.reshape(B * C, N), n_fft=self.n_fft, hop_length=self.n_overlap, win_length=self.n_fft, window=self.window, return_complex=True)
        Fq, TF = X.shape[-2:]
        X = X.reshape(B, C, Fq, TF).permute(0, 2, 3, 1)  # [B,F,T,C]
        scale = X[..., self.ref_channel].abs().mean(dim=2)  # [B,F]: mean magnitude of the reference channel per frequency
        feats = torch.view_as_real(X / (scale[:, :, None, None] + 1e-8)).reshape(B, Fq, TF, 2 * C)
        out = self.arch(feats).reshape(B, Fq, TF, self.n_speaker, 2)
        Y = torch.view_as_complex(out.float().contiguous()) * scale[:, :, None, None]  # [B,F,T,S]
        Y = Y.permute(0, 3, 1, 2).reshape(B * self.n_speaker, Fq, TF)
        if native_io:
            y = io.istft(Y, N)  # (autograd.Function: its backward is the iSTFT adjoint kernel)
        else:
            y = torch.istft(Y, n_fft=self.n_fft, hop_length=self.n_overlap, win_length=self.n_fft, window=self.window, length=N)
        return y.reshape(B, self.n_speaker, N)
