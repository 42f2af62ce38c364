"""ORACLE — test infrastructure only (see oracle/spatialnet_ref.py for the rules).

CPU restatement of the reference's signal I/O, loss and training-step glue:
  models/io/stft.py (STFT.stft / istft), models/io/norm.py (Norm 'frequency', online),
  models/io/loss.py (Loss(neg_si_sdr, pit=True)) and SharedTrainer.py:104-149 (TrainModule.forward /
  training_step), general_steps.py:243-271 + configs/SpatialNet.yaml:3-4,44 (clip 5 + Adam).

torchmetrics (requirements.txt:2, unpinned, NOT installed here) supplies si_sdr / pit in the
reference; their published definitions are restated below.  The reference has no test that pins
them, so this boundary is "parity unpinned" beyond the closed forms and the known-answer vectors in
tests/test_signal_loss_optim.py::test_sisdr_known_answers.
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, Tuple

import torch
from torch import Tensor

from . import spatialnet_ref as net


def hann(n_fft: int, kind: str = "hann_window", dtype=torch.float32) -> Tensor:
    w = torch.hann_window(n_fft, dtype=torch.float64)  # periodic (stft.py:27-33)
    if kind == "sqrt_hann_window":
        w = w.sqrt()
    return w.to(dtype)


def stft(x: Tensor, n_fft: int = 256, hop: int = 128, win: str = "hann_window") -> Tensor:
    """STFT.stft (stft.py:49-66): [..., N] -> complex [..., F, T]; centre=True, reflect pad, one-sided."""
    shape = list(x.shape)
    X = torch.stft(x.reshape(-1, shape[-1]), n_fft=n_fft, hop_length=hop, win_length=n_fft, window=hann(n_fft, win, x.dtype), return_complex=True)
    return X.reshape(shape[:-1] + list(X.shape[-2:]))


def istft(X: Tensor, length: int, n_fft: int = 256, hop: int = 128, win: str = "hann_window") -> Tensor:
    """STFT.istft (stft.py:68-97): one torch.istft call per item, like the reference's loop."""
    shape = list(X.shape)
    Xf = X.reshape(-1, *shape[-2:])
    w = hann(n_fft, win, Xf.real.dtype)
    xs = [torch.istft(Xf[b], n_fft=n_fft, hop_length=hop, win_length=n_fft, window=w, length=length) for b in range(Xf.shape[0])]
    return torch.stack(xs, 0).reshape(shape[:-2] + [length])


def norm_frequency_online(X: Tensor, ref_channel: int, eps: float = 1e-6) -> Tuple[Tensor, Tensor]:
    """Norm('frequency', online=True).norm (norm.py:77-81,94): XrMM = |X_ref| + eps per T-F bin.  X [B,C,F,T] complex."""
    XrMM = X[:, [ref_channel]].abs() + eps
    return X / XrMM, XrMM


def to_real_layout(X: Tensor) -> Tensor:
    """[B,C,F,T] complex -> [B,F,T,2C] real (SharedTrainer.py:116-117)."""
    B, C, F, T = X.shape
    return torch.view_as_real(X.permute(0, 2, 3, 1).contiguous()).reshape(B, F, T, 2 * C)


def from_real_layout(out: Tensor) -> Tensor:
    """[B,F,T,2S] real -> [B,S,F,T] complex (SharedTrainer.py:121-123)."""
    B, F, T, S2 = out.shape
    return torch.view_as_complex(out.reshape(B, F, T, S2 // 2, 2).contiguous()).permute(0, 3, 1, 2)


def si_sdr(p: Tensor, t: Tensor) -> Tensor:
    """torchmetrics.functional.audio.scale_invariant_signal_distortion_ratio(zero_mean=False) over the last dim."""
    eps = torch.finfo(p.dtype).eps
    alpha = ((p * t).sum(-1, keepdim=True) + eps) / ((t * t).sum(-1, keepdim=True) + eps)
    ts = alpha * t
    noise = ts - p
    return 10 * torch.log10(((ts * ts).sum(-1) + eps) / ((noise * noise).sum(-1) + eps))


def neg_si_sdr(p: Tensor, t: Tensor) -> Tensor:
    """models/io/loss.py:21-29: -mean over speakers -> [B]."""
    return -si_sdr(p, t).reshape(t.shape[0], -1).mean(1)


def pit_neg_si_sdr(p: Tensor, t: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Loss(neg_si_sdr, pit=True).forward (loss.py:107-118) with torchmetrics pit(mode='permutation-wise',
    eval_func='min'): evaluate metric(preds[:, perm], target) for every permutation (itertools order), take
    the min.  Returns (mean loss, per-item loss [B], perm [B,S])."""
    B, S = p.shape[:2]
    perms = list(itertools.permutations(range(S)))
    vals = torch.stack([neg_si_sdr(p[:, list(pm)], t) for pm in perms], 1)  # [B, S!]
    best, idx = vals.min(1)
    perm = torch.tensor(perms, dtype=torch.long)[idx]
    return best.mean(), best, perm


def train_forward(x: Tensor, yr: Tensor, p: Dict[str, Tensor], num_layers: int, ref_channel: int = 0, n_fft: int = 256, hop: int = 128):
    """TrainModule.forward + training_step loss (SharedTrainer.py:104-149) for configs/SpatialNet.yaml:
    x [B,C,N] mixture, yr [B,S,N] reference-channel targets -> (loss, yr_hat [B,S,N], network output [B,F,T,2S])."""
    N = x.shape[-1]
    X = stft(x, n_fft, hop)
    Xn, XrMM = norm_frequency_online(X, ref_channel)
    out = net.spatialnet(to_real_layout(Xn), p, num_layers)
    Yr_hat = from_real_layout(out) * XrMM
    yr_hat = istft(Yr_hat, N, n_fft, hop)
    loss, _, _ = pit_neg_si_sdr(yr_hat, yr)
    return loss, yr_hat, out
