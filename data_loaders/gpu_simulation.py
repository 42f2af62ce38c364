"""On-GPU simulation of multichannel noisy-reverberant mixtures: the per-item numpy pipeline of the
reference's dataset modules, batched on the device so that the loader keeps up with the HIP training step (hundreds of utterances/s
per GPU; the reference's 10 CPU workers produce far fewer):
    dry sources  --RIR convolution (FFT), aligned to the direct path of the reference channel-->  reverberant images / targets
    --relative scaling of speaker 2 to a sampled SIR-->  mixture  --spatially diffuse noise at a sampled SNR-->  peak scaling 0.9
following data_loaders/sms_wsj_plus.py:157-220, utils/mix.py:122-134 (convolve), :328-346 (energy ratio), utils/diffuse_noise.py:19-93
(spherically isotropic coherence sinc(w d / c), mixing matrices from its eigendecomposition, STFT-domain mixing of independent
noises).  Everything is torch (hipFFT / rocSOLVER through torch-ROCm) and device-agnostic: the CPU tests check it against
scipy and, when the reference tree is present, against the reference's own functions.
The corpora (WSJ0, measured/simulated RIR sets) are not available here: `SimulatedRoomDataModule` draws band-limited random
sources and synthetic exponentially decaying RIRs; `mix_batch` itself takes any sources / RIRs."""
import math
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from data_loaders.synthetic import rank_strided_indices


def fft_convolve(x: Tensor, h: Tensor) -> Tensor:
    """full linear convolution along the last axis (scipy.signal.fftconvolve(mode='full')); leading dims broadcast"""
    n = x.shape[-1] + h.shape[-1] - 1
    nfft = 1 << (n - 1).bit_length()
    return torch.fft.irfft(torch.fft.rfft(x, nfft) * torch.fft.rfft(h, nfft), nfft)[..., :n]


def convolve_aligned(wav: Tensor, rir: Tensor, rir_target: Optional[Tensor] = None, ref_channel: int = 0) -> Tuple[Tensor, Tensor]:
    """wav [B,S,N], rir [B,S,M,L] -> (reverberant image, target image) [B,S,M,N], both cut from the direct-path delay of the
    reference channel of `rir` (utils/mix.py:122-134, align=True)"""
    B, S, N = wav.shape
    full = fft_convolve(wav[:, :, None, :], rir)
    tgt = full if rir_target is None else fft_convolve(wav[:, :, None, :], rir_target)
    delay = rir[:, :, ref_channel].argmax(-1)  # [B,S]
    idx = delay[:, :, None, None] + torch.arange(N, device=wav.device)[None, None, None, :]
    idx = idx.expand(B, S, rir.shape[2], N)
    return full.gather(-1, idx), tgt.gather(-1, idx)


def energy_ratio_coeff(a: Tensor, b: Tensor, target_db: Tensor) -> Tensor:
    """factor for `b` such that 10 log10(mean(a^2) / mean((coeff b)^2)) = target_db; a, b [B,...], target_db [B] (mix.py:328-346)"""
    ea = a.reshape(a.shape[0], -1).pow(2).mean(1)
    eb = b.reshape(b.shape[0], -1).pow(2).mean(1)
    return torch.sqrt(ea / eb * torch.pow(10.0, -target_db / 10))


def diffuse_mixing_matrices(pos_mics: Tensor, fs: int, nfft: int = 256, c: float = 343.0) -> Tuple[Tensor, Tensor]:
    """spherically isotropic noise field: coherence DSC[m,n,k] = sinc(w_k d_mn / c) and one mixing matrix per frequency with
    C^H C = DSC (eigendecomposition; frequency 0 is left empty like in diffuse_noise.py:40-59) -> (DSC [M,M,F], Cs [F,M,M] complex)"""
    M, Fq = pos_mics.shape[0], nfft // 2 + 1
    w = 2 * math.pi * fs * torch.arange(Fq, dtype=torch.float64, device=pos_mics.device) / nfft
    dist = (pos_mics[:, None, :] - pos_mics[None, :, :]).double().norm(dim=-1)
    dsc = torch.sinc(w[None, None, :] * dist[:, :, None] / (c * math.pi))  # torch.sinc(x) = sin(pi x) / (pi x), as numpy's
    ev, V = torch.linalg.eigh(dsc.permute(2, 0, 1))  # symmetric PSD: [F,M], [F,M,M]
    Cs = (V.transpose(1, 2) * ev.clamp(min=0).sqrt()[:, :, None]).to(torch.complex128)
    Cs[0] = 0
    return dsc, Cs


def _stft_scipy(x: Tensor, nfft: int) -> Tensor:
    """scipy.signal.stft(window='hann', nperseg=nfft, noverlap=0.75 nfft, boundary='zeros', padded=True): [..., L] -> [..., F, T]"""
    hop = nfft // 4
    x = torch.nn.functional.pad(x, (nfft // 2, nfft // 2))
    extra = (-(x.shape[-1] - nfft)) % hop
    x = torch.nn.functional.pad(x, (0, extra))
    win = torch.hann_window(nfft, periodic=True, dtype=x.dtype, device=x.device)
    fr = x.unfold(-1, nfft, hop) * win
    return torch.fft.rfft(fr, nfft).transpose(-1, -2) / win.sum()


def _istft_scipy(X: Tensor, nfft: int) -> Tensor:
    """inverse of _stft_scipy (overlap-add with the hann window, window-envelope normalisation, boundary trimmed)"""
    hop = nfft // 4
    win = torch.hann_window(nfft, periodic=True, dtype=X.real.dtype, device=X.device)
    fr = torch.fft.irfft(X.transpose(-1, -2) * win.sum(), nfft) * win  # [..., T, nfft]
    T = fr.shape[-2]
    n = nfft + hop * (T - 1)
    lead = fr.shape[:-2]
    out = torch.nn.functional.fold(fr.reshape(-1, T, nfft).transpose(1, 2), (1, n), (1, nfft), stride=(1, hop)).reshape(*lead, n)
    env = torch.nn.functional.fold((win * win).expand(1, T, nfft).transpose(1, 2), (1, n), (1, nfft), stride=(1, hop)).reshape(n)
    return (out / env.clamp(min=1e-10))[..., nfft // 2: n - nfft // 2]


def gen_diffuse_noise(noise: Tensor, L: int, Cs: Tensor, nfft: int = 256) -> Tensor:
    """M mutually independent noises [..., M, >= L] -> diffuse noise [..., M, L] with the coherence the matrices `Cs` encode
    (diffuse_noise.py:64-93: zero-mean, STFT, X[n] = sum_m conj(C[f,m,n]) N[m], inverse STFT)"""
    noise = noise[..., :L] - noise[..., :L].mean(-1, keepdim=True)
    N = _stft_scipy(noise, nfft)  # [..., M, F, T]
    X = torch.einsum("fmn,...mft->...nft", Cs.conj().to(N.dtype), N)
    return _istft_scipy(X, nfft)[..., :L]


def mix_batch(cleans: Tensor, rir: Tensor, Cs: Tensor, sir_db: Optional[Tensor], snr_db: Tensor, gen: Optional[torch.Generator], rir_target: Optional[Tensor] = None,
              nfft: int = 256, white: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Dict[str, Tensor]]:
    """cleans [B,S,N] dry sources, rir [B,S,M,L] -> (mix [B,M,N], targets [B,S,M,N], paras): steps 5-7 of SmsWsjPlusDataset.__getitem__
    (full overlap) for a whole batch on the device of `cleans`"""
    B, S, N = cleans.shape
    M = rir.shape[2]
    rvbt, tgt = convolve_aligned(cleans, rir, rir_target)
    if sir_db is not None and S == 2:  # speaker 2 relative to speaker 1
        coeff = energy_ratio_coeff(rvbt[:, 0], rvbt[:, 1], sir_db)
        scale = torch.stack([torch.ones_like(coeff), coeff], 1)[:, :, None, None]
        rvbt, tgt = rvbt * scale, tgt * scale
    mix = rvbt.sum(1)
    if white is None:
        white = torch.randn(B, M, N, generator=gen, device=cleans.device, dtype=cleans.dtype)
    noise = gen_diffuse_noise(white, N, Cs.to(cleans.device), nfft)
    noise = noise * energy_ratio_coeff(mix, noise, snr_db)[:, None, None]
    snr_real = 10 * torch.log10(mix.pow(2).sum((1, 2)) / noise.pow(2).sum((1, 2)))
    mix = mix + noise
    peak = torch.maximum(mix.abs().amax((1, 2)), tgt.abs().amax((1, 2, 3)))
    s = 0.9 / peak
    return mix * s[:, None, None], tgt * s[:, None, None, None], {"snr": snr_real, "scale": s}


class SimulatedRoomDataModule:
    """The batch contract of the reference's data modules (x [B,C,N], ys [B,Spk,C,N], paras), produced ON THE TRAINING DEVICE:
    synthetic band-limited sources, synthetic RIRs (direct path + exponentially decaying diffuse tail, RT60 sampled per item, mic
    delays from a circular array geometry), SIR / SNR sampled like configs/datasets/sms_wsj_plus.yaml.  Items are addressed by
    (index, seed) and sharded rank-strided like MyDistributedSampler."""

    def __init__(self, batch_size: List[int] = (2, 2), num_samples: List[int] = (64, 8, 8), audio_time_len: List[float] = (4.0, 4.0, 4.0),
                 num_channels: int = 6, num_speakers: int = 2, sample_rate: int = 8000, sir: Tuple[float, float] = (-5.0, 5.0),
                 snr: Tuple[float, float] = (0.0, 20.0), rt60: Tuple[float, float] = (0.2, 0.6), array_radius: float = 0.1, seeds: List[int] = (0, 1, 2),
                 device: Optional[str] = None):
        self.batch_size, self.num_samples, self.audio_time_len = list(batch_size), list(num_samples), list(audio_time_len)
        self.C, self.S, self.sr, self.sir, self.snr, self.rt60, self.seeds = num_channels, num_speakers, sample_rate, sir, snr, rt60, list(seeds)
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        ang = torch.arange(num_channels) * (2 * math.pi / num_channels)
        self.pos_mics = torch.stack([array_radius * torch.cos(ang), array_radius * torch.sin(ang), torch.zeros(num_channels)], 1)
        self.Cs = diffuse_mixing_matrices(self.pos_mics, sample_rate)[1].to(torch.complex64).to(self.device)

    def _rirs(self, B: int, gen: torch.Generator) -> Tensor:
        """[B,S,M,L]: unit direct path at a per-(speaker, mic) delay from a random far-field direction + decaying Gaussian tail"""
        dev, L = self.device, int(0.4 * self.sr)
        rt = self.rt60[0] + (self.rt60[1] - self.rt60[0]) * torch.rand(B, 1, 1, 1, generator=gen, device=dev)
        t = torch.arange(L, device=dev) / self.sr
        tail = torch.randn(B, self.S, self.C, L, generator=gen, device=dev) * torch.exp(-6.9 * t / rt) * 0.3
        az = 2 * math.pi * torch.rand(B, self.S, generator=gen, device=dev)
        direction = torch.stack([torch.cos(az), torch.sin(az), torch.zeros_like(az)], -1)  # [B,S,3]
        delay = (16 + (-(direction @ self.pos_mics.to(dev).T) / 343.0 * self.sr)).round().long().clamp(0, L - 1)  # [B,S,M]
        tail.masked_fill_(torch.arange(L, device=dev)[None, None, None, :] < delay[..., None], 0.0)
        return tail.scatter(-1, delay[..., None], 1.0)

    def batches(self, stage: int, rank: int = 0, world: int = 1, epoch: int = 0):
        N = int(self.audio_time_len[stage] * self.sr)
        bs = self.batch_size[min(stage, len(self.batch_size) - 1)]
        items = rank_strided_indices(self.num_samples[stage], rank, world, epoch, self.seeds[stage], shuffle=stage == 0)
        k = torch.hann_window(33, device=self.device)
        for i in range(0, len(items) - bs + 1, bs):
            chunk = items[i:i + bs]
            gen = torch.Generator(device=self.device).manual_seed(int(chunk[0][1]) * 1000003 + int(chunk[0][0]))
            src = torch.randn(bs * self.S, 1, N, generator=gen, device=self.device)
            src = torch.nn.functional.conv1d(src, (k / k.sum())[None, None], padding=16).reshape(bs, self.S, N) * 3.0
            sir = self.sir[0] + (self.sir[1] - self.sir[0]) * torch.rand(bs, generator=gen, device=self.device)
            snr = self.snr[0] + (self.snr[1] - self.snr[0]) * torch.rand(bs, generator=gen, device=self.device)
            mix, tgt, paras = mix_batch(src, self._rirs(bs, gen), self.Cs, sir if self.S == 2 else None, snr, gen)
            yield mix, tgt, [{"index": ix, "seed": sd, "sample_rate": self.sr, "snr": float(paras["snr"][j]), "sir": float(sir[j])}
                             for j, (ix, sd) in enumerate(chunk)]
