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


def _step_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nbss_amd._lib import NBSS_F32, Lib
    from nbss_amd.build import build_emu
    from nbss_amd.engine import SpatialNetEngine, TrainStep
    lib = Lib(build_emu())
    g = torch.Generator().manual_seed(500 + rank)  # rank-dependent data
    xin = torch.randn(1, 9, 21, 4, generator=g)
    dout = torch.randn(1, 9, 21, 4, generator=g)
    out = {}
    for mode in ("bucketed", "single"):
        torch.manual_seed(0)  # identical parameters on every rank and in both modes
        eng = SpatialNetEngine(lib, "cpu", dim_input=4, dim_output=4, num_freqs=9, num_layers=3, dtype=NBSS_F32)
        eng.params.copy_(torch.randn_like(eng.params) * 0.05)
        eng.version += 1
        ts = TrainStep(eng, lr=1e-2, clip=5.0, bucketed=mode == "bucketed")
        assert ts.world == world and len(eng.grad_buckets()) == 3
        eng.forward(xin, train=True)
        ts.backward_and_update(xin, dout)  # network backward (+ per-layer async all-reduce) + clip + Adam
        out[mode] = {"params": eng.params.clone(), "norm": float(ts.scratch[0])}
    torch.save(out, f"{tmp}/s{rank}.pt")
    dist.destroy_process_group()


def test_overlapped_bucket_allreduce_equals_single_allreduce(tmp_path):
    """per-layer gradient buckets reduced during backward (nbss_spatialnet_bwd_range + async all-reduce) give the same update as
    one all-reduce after backward, and replicas stay bit-identical (atomicAdd order makes the two MODES agree to rounding only)"""
    world, port = 2, 31500 + os.getpid() % 2000
    mp.spawn(_step_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"s{r}.pt") for r in range(world)]
    for mode in ("bucketed", "single"):
        assert torch.equal(outs[0][mode]["params"], outs[1][mode]["params"])
    a, b = outs[0]["bucketed"], outs[0]["single"]
    assert abs(a["norm"] - b["norm"]) < 1e-5 * b["norm"] and b["norm"] > 0
    assert float((a["params"] - b["params"]).norm() / b["params"].norm()) < 1e-6


def _replica_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nbss_amd._lib import NBSS_F32, Lib
    from nbss_amd.build import build_emu
    from nbss_amd.engine import SpatialNetEngine, TrainStep
    lib = Lib(build_emu())
    torch.manual_seed(10 + rank)  # DIFFERENT parameters / optimizer state per rank: what a resume on rank 0 only looks like
    eng = SpatialNetEngine(lib, "cpu", dim_input=4, dim_output=4, num_freqs=9, num_layers=1, dtype=NBSS_F32)
    eng.params.copy_(torch.randn_like(eng.params) * 0.1)
    ts = TrainStep(eng, lr=1e-2 * (rank + 1), clip=0.5)
    ts.m.copy_(torch.randn_like(ts.m))
    ts.v.copy_(torch.rand_like(ts.v))
    ts.step_count = 5 + rank
    diverged = False
    try:
        ts.check_replicas()
    except RuntimeError:
        diverged = True
    ts.sync_replicas()
    spread = ts.check_replicas()
    torch.save({"params": eng.params.clone(), "m": ts.m.clone(), "v": ts.v.clone(), "step": ts.step_count, "lr": ts.lr, "diverged": diverged, "spread": spread},
               f"{tmp}/q{rank}.pt")
    dist.destroy_process_group()


def test_replica_broadcast_and_checksum(tmp_path):
    """TrainStep.sync_replicas makes rank 0's parameters, Adam moments, step count and learning rate everyone's (DDP's init broadcast,
    SURVEY.md §2.4); check_replicas raises while they differ and returns 0.0 afterwards"""
    world, port = 2, 33500 + os.getpid() % 2000
    mp.spawn(_replica_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(tmp_path / f"q{r}.pt") for r in range(world))
    assert a["diverged"] and b["diverged"] and a["spread"] == 0.0 and b["spread"] == 0.0
    for k in ("params", "m", "v"):
        assert torch.equal(a[k], b[k])
    assert a["step"] == b["step"] == 5 and a["lr"] == b["lr"] == 1e-2
