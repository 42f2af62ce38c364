"""Data-parallel logic on CPU with the gloo backend, world_size 2 (the GPU path uses RCCL through the same calls):
rank-strided sharding (MyDistributedSampler semantics) and the one-collective gradient exchange of TrainStep — all-reduce
(SUM) of the flat fp32 gradient, mean folded into the clip+Adam kernel — must equal a single-process update with the averaged
gradient."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from data_loaders.synthetic import rank_strided_indices


def test_rank_strided_sharding_covers_everything_once():
    n, world = 37, 4
    shards = [rank_strided_indices(n, r, world, epoch=3, seed=11) for r in range(world)]
    assert len({len(s) for s in shards}) == 1  # equal length on every rank (padded by wrap-around)
    flat = [ix for s in shards for ix, _ in s]
    assert set(flat) == set(range(n)) and len(flat) == (n + world - 1) // world * world
    assert shards[0] != rank_strided_indices(n, 0, world, epoch=4, seed=11)  # reshuffled per epoch
    assert shards[1] == rank_strided_indices(n, 1, world, epoch=3, seed=11)  # deterministic


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nbss_amd._lib import NBSS_F32, Lib
    from nbss_amd.build import build_emu
    from nbss_amd.engine import SpatialNetEngine, TrainStep
    lib = Lib(build_emu())
    torch.manual_seed(0)  # identical parameters on every rank
    eng = SpatialNetEngine(lib, "cpu", dim_input=4, dim_output=4, num_freqs=129, num_layers=1, dtype=NBSS_F32)
    eng.params.copy_(torch.randn_like(eng.params) * 0.1)
    ts = TrainStep(eng, lr=1e-2, clip=0.5)
    assert ts.world == world
    g = torch.Generator().manual_seed(100 + rank)  # rank-dependent "local" gradient
    local = torch.randn(eng.params.shape, generator=g)
    eng.grads.copy_(local)
    ts.apply_gradients()
    torch.save({"params": eng.params.clone(), "local": local, "norm": ts.scratch[0].clone()}, f"{tmp}/r{rank}.pt")
    dist.destroy_process_group()


def test_gradient_allreduce_matches_single_process_mean(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert torch.equal(outs[0]["params"], outs[1]["params"])  # replicas stay bit-identical
    mean = sum(o["local"] for o in outs) / world
    torch.manual_seed(0)
    p = (torch.randn_like(mean) * 0.1).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-2)
    p.grad = mean.clone()
    norm = torch.nn.utils.clip_grad_norm_([p], 0.5)
    opt.step()
    assert abs(float(outs[0]["norm"]) - float(norm)) < 1e-4 * float(norm)
    assert float((outs[0]["params"] - p.detach()).norm() / p.detach().norm()) < 1e-6
