"""Synthetic multichannel mixtures with the reference's batch contract (data_loaders/sms_wsj_plus.py:248,
utils/collate_func.py:8-16): x [B,C,N] fp32, ys [B,Spk,C,N] fp32, paras list[dict]; items are addressed by
(index, seed) and sharded rank-strided like MyDistributedSampler (data_loaders/utils/my_distributed_sampler.py:64-79).
The real corpora (WSJ0, RIRs) are not available; the bench metric is defined on synthetic data."""
from typing import List, Tuple

import torch


def rank_strided_indices(n: int, rank: int, world: int, epoch: int, seed: int, shuffle: bool = True) -> List[Tuple[int, int]]:
    """(index, seed) pairs of this rank: shuffle with (seed + epoch), pad by wrap-around to a multiple of world, take rank::world."""
    g = torch.Generator().manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist() if shuffle else list(range(n))
    seeds = torch.randint(0, 2**31 - 1, (n,), generator=g).tolist()
    pairs = list(zip(idx, seeds))
    total = (n + world - 1) // world * world
    pairs += pairs[: total - n]
    return pairs[rank:total:world]


class SyntheticDataModule:
    def __init__(self, batch_size: List[int] = (2, 2), num_samples: List[int] = (64, 8, 8), audio_time_len: List[float] = (4.0, 4.0, 4.0),
                 num_channels: int = 6, num_speakers: int = 2, sample_rate: int = 8000, seeds: List[int] = (0, 1, 2)):
        self.batch_size, self.num_samples, self.audio_time_len = list(batch_size), list(num_samples), list(audio_time_len)
        self.C, self.S, self.sr, self.seeds = num_channels, num_speakers, sample_rate, list(seeds)

    def _item(self, index: int, seed: int, N: int):
        g = torch.Generator().manual_seed(seed * 1000003 + index)
        src = torch.randn(self.S, N, generator=g)
        k = torch.hann_window(33)[None, None]
        src = torch.nn.functional.conv1d(src[:, None], k / k.sum(), padding=16)[:, 0] * 3.0
        gains = 0.5 + torch.rand(self.C, self.S, generator=g)
        ys = gains.t()[:, :, None] * src[:, None, :]                    # [S,C,N]
        x = ys.sum(0) + 0.01 * torch.randn(self.C, N, generator=g)      # [C,N]
        return x, ys, {"index": index, "seed": seed, "sample_rate": self.sr}

    def batches(self, stage: int, rank: int = 0, world: int = 1, epoch: int = 0):
        """stage 0 train / 1 val / 2 test"""
        N = int(self.audio_time_len[stage] * self.sr)
        bs = self.batch_size[min(stage, len(self.batch_size) - 1)]
        items = rank_strided_indices(self.num_samples[stage], rank, world, epoch, self.seeds[stage], shuffle=stage == 0)
        for i in range(0, len(items) - bs + 1, bs):
            its = [self._item(ix, sd, N) for ix, sd in items[i:i + bs]]
            yield torch.stack([a for a, _, _ in its]), torch.stack([b for _, b, _ in its]), [c for _, _, c in its]
