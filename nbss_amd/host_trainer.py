"""Generic trainer loop behind `SharedTrainer.py fit` for everything that is NOT the fused SpatialNet step: the narrow-band archs
(NB-BLSTM / NBC / NBC2) and OnlineSpatialNet on a HIP device (PyTorch-ROCm compute; BASELINE configs 4, 5) or on the host
(`trainer.accelerator=cpu`; BASELINE config 1 is exactly that: NB-BLSTM, 2 channels, 1-s utterances, CPU).
It drives TrainModule.training_step with torch.optim, clip_grad_norm_ and the configured scheduler — the reference's Lightning loop
(general_steps.py:243-271, configs/*.yaml trainer section) without Lightning.  Under torchrun (WORLD_SIZE > 1) the samples are
rank-strided (my_distributed_sampler.py:64-79), gradients are averaged with one all-reduce of the flattened gradient per step and
rank 0 writes the checkpoint.  models.arch.SpatialNet has no host path and raises."""
from __future__ import annotations

import json
import os
import time
from typing import Any, Dict

import torch


def fit_generic(cfg: dict, build_module, instantiate) -> Dict[str, Any]:
    from models.arch.SpatialNet import SpatialNet
    tr = cfg.get("trainer", {})
    use_gpu = tr.get("accelerator", "gpu") != "cpu" and torch.cuda.is_available()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))) if use_gpu else torch.device("cpu")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if use_gpu:
        torch.cuda.set_device(dev)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl" if use_gpu else "gloo")
    torch.manual_seed(int(cfg.get("seed_everything", 2)))  # same seed on every rank: identical initial replicas
    module = build_module(cfg)
    if isinstance(module.arch, SpatialNet) and dev.type != "cuda":
        raise RuntimeError("SharedTrainer fit: models.arch.SpatialNet runs on MI355X HIP kernels only (no CPU path)")
    module = module.to(dev)
    module.precision = str(tr.get("precision", "32"))
    data = instantiate(cfg["data"]) if "data" in cfg else None
    if data is None:
        from data_loaders.synthetic import SyntheticDataModule
        data = SyntheticDataModule()
    first_epoch = 0
    opt, sched = module.configure_optimizers()
    if cfg.get("ckpt_path"):
        ck = torch.load(cfg["ckpt_path"], map_location="cpu", weights_only=False)
        sd = {k.replace("_orig_mod.", "").removeprefix("arch."): v for k, v in ck["state_dict"].items() if not k.endswith("stft.window")}
        module.arch.load_state_dict(sd, strict=True)
        if ck.get("optimizer_states"):
            opt.load_state_dict(ck["optimizer_states"][0])
        else:
            print(f"[SharedTrainer] {cfg['ckpt_path']} has no optimizer state: moments, step count and learning rate start fresh", flush=True)
        if sched is not None and ck.get("lr_schedulers"):
            try:
                sched.load_state_dict(ck["lr_schedulers"][0])
            except Exception as e:  # a scheduler state written by another trainer / scheduler class
                print(f"[SharedTrainer] lr scheduler state not restored ({type(e).__name__}: {e})", flush=True)
        first_epoch = int(ck.get("epoch", -1)) + 1
    clip = float(tr.get("gradient_clip_val") or 0.0)
    plateau = isinstance(sched, torch.optim.lr_scheduler.ReduceLROnPlateau)
    ckpt_dir = tr.get("default_root_dir")
    log, step = [], 0
    for epoch in range(first_epoch, int(tr.get("max_epochs", 1))):
        t0, n, tot = time.time(), 0, 0.0
        module.train()
        for x, ys, paras in data.batches(0, rank, world, epoch):
            opt.zero_grad(set_to_none=True)
            loss = module.training_step((x.to(dev), ys.to(dev), paras))
            loss.backward()
            if world > 1:
                grads = [p.grad for p in module.parameters() if p.grad is not None]
                flat = torch.cat([g.reshape(-1) for g in grads])
                torch.distributed.all_reduce(flat)
                flat /= world
                off = 0
                for g in grads:
                    g.copy_(flat[off:off + g.numel()].view_as(g))
                    off += g.numel()
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(module.parameters(), clip)
            opt.step()
            tot += float(loss.detach())
            n += 1
            step += 1
        module.eval()
        vtot, vn = 0.0, 0
        with torch.no_grad():
            for x, ys, paras in data.batches(1, 0, 1, 0):
                vtot += float(module.training_step((x.to(dev), ys.to(dev), paras)))
                vn += 1
        val = vtot / max(vn, 1)
        if sched is not None:
            sched.step(val) if plateau else sched.step()
        rec = {"epoch": epoch, "train/neg_si_sdr": tot / max(n, 1), "val/neg_si_sdr": val, "steps": n, "lr": opt.param_groups[0]["lr"],
               "sec": time.time() - t0, "device": str(dev)}
        log.append(rec)
        if rank == 0:
            print(json.dumps(rec), flush=True)
        if ckpt_dir and rank == 0:
            os.makedirs(os.path.join(ckpt_dir, "checkpoints"), exist_ok=True)
            sd = {"arch." + k: v.detach().cpu().clone() for k, v in module.arch.state_dict().items()}
            sd["stft.window"] = module.stft.window.detach().cpu().clone()
            torch.save({"epoch": epoch, "global_step": step, "pytorch-lightning_version": "2.0.0", "state_dict": sd,
                        "optimizer_states": [opt.state_dict()], "lr_schedulers": [sched.state_dict()] if sched is not None else []},
                       os.path.join(ckpt_dir, "checkpoints", "last.ckpt"))
    if world > 1:
        torch.distributed.barrier()
    return {"log": log, "module": module}
