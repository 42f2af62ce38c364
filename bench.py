#!/usr/bin/env python
"""bench.py — utterances/sec of a full SpatialNet-small bf16-mixed TRAINING step on MI355X.

Workload (BASELINE.json configs[1]): synthetic 4-s, 6-channel, 8-kHz mixtures
(x[B,6,32000], 2 speakers), n_fft 256 / hop 128 -> F=129, T=251; one step = STFT + per-bin norm ->
SpatialNet-small (8 layers, H=96) forward -> inorm + iSTFT -> uPIT neg-SI-SDR -> full backward ->
[gradient all-reduce over RCCL] -> clip(5) + Adam(1e-3) -> re-pack of the MFMA weight fragments.
Inputs are resident in HBM before the timed region.  One process per GPU (torchrun sets RANK / LOCAL_RANK /
WORLD_SIZE), weak scaling: the per-GPU batch is fixed.

Prints ONE JSON line (rank 0) with the contract keys plus
  roofline     : dominant kernel (largest share of GPU time), algorithmic HBM bytes per launch / its average
                 duration measured with HIP events on the launch stream during the timed steps, vs 8 TB/s
  cpu_baseline : the oracle (a CPU restatement of the reference's fp32 training step) timed on this host
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

S_BYTES_BF16 = 129 * 251 * 96 * 2  # one utterance's residual stream (S)
HBM_PEAK = 8.0e12


def synth_batch(B, C, S, N, seed, device):
    """two random 'speech-like' sources (low-passed noise bursts) mixed into C channels with random gains/delays + noise"""
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(B, S, N, generator=g)
    k = torch.hann_window(33)[None, None]
    src = torch.nn.functional.conv1d(src.reshape(B * S, 1, N), k / k.sum(), padding=16).reshape(B, S, N)
    env = (torch.rand(B, S, N // 800 + 1, generator=g) > 0.3).float().repeat_interleave(800, -1)[..., :N]
    src = src * env * 3.0
    gains = 0.5 + torch.rand(B, C, S, generator=g)
    mix = torch.einsum("bcs,bsn->bcn", gains, src) + 0.01 * torch.randn(B, C, N, generator=g)
    yr = src * gains[:, 0, :, None]  # targets at the reference channel
    return mix.to(device), yr.contiguous().to(device)


def profile_read(lib):
    n = lib.nbss_profile_kernels()
    ms = (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    lib.nbss_profile_read(ms, cnt)
    return {lib.nbss_profile_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}


def algorithmic_bytes(name, B):
    """per-launch algorithmic HBM bytes (each residual sub-block reads + writes the stream once in
    forward (2S) and reads x, dy, writes dx in backward (3S); S = 6.2 MB per utterance in bf16)"""
    S = S_BYTES_BF16 * B
    if name.endswith("_fwd") and name.split("_")[0] in ("fconv", "full", "mhsa", "tconvffn"):
        return 2 * S
    if name.endswith("_bwd") and name.split("_")[0] in ("fconv", "full", "mhsa", "tconvffn"):
        return 3 * S
    return None


def _cpu_train_step_fn(B, use_reference):
    """one fp32 SpatialNet-small training step on the host (STFT .. clip + Adam), reference semantics.  use_reference: the network,
    STFT and Norm are the reference's OWN modules imported from /root/reference (present in the build container only); the uPIT
    neg-SI-SDR loss is the oracle's restatement either way (the reference takes it from torchmetrics, which is not installed)."""
    from oracle import io_ref
    from oracle import spatialnet_ref as ref
    torch.manual_seed(2)
    x, yr = synth_batch(B, 6, 2, 32000, 99, "cpu")
    if use_reference:
        import importlib
        saved = {k: sys.modules.pop(k) for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]}
        sys.path.insert(0, "/root/reference")
        try:
            RefNet = importlib.import_module("models.arch.SpatialNet").SpatialNet
            RefSTFT = importlib.import_module("models.io.stft").STFT
            RefNorm = importlib.import_module("models.io.norm").Norm
        finally:
            sys.path.remove("/root/reference")
            for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
                del sys.modules[k]
            sys.modules.update(saved)
        net = RefNet(dim_input=12, dim_output=4, num_layers=8, encoder_kernel_size=5, dim_hidden=96, dim_ffn=192, num_heads=4, dropout=(0, 0, 0),
                     kernel_size=(5, 3), conv_groups=(8, 8), norms=("LN", "LN", "GN", "LN", "LN", "LN"), dim_squeeze=8, num_freqs=129, full_share=0)
        stft, norm = RefSTFT(n_fft=256, n_hop=128), RefNorm(mode="frequency")
        leaves = list(net.parameters())
        opt = torch.optim.Adam(leaves, lr=1e-3)

        def step():
            opt.zero_grad(set_to_none=True)
            X, n = stft.stft(x)
            Bq, C, F, T = X.shape
            X, (Xr, XrMM) = norm.norm(X, ref_channel=0)
            out = net(torch.view_as_real(X.permute(0, 2, 3, 1)).reshape(Bq, F, T, -1))
            out = torch.view_as_complex(out.float().reshape(Bq, F, T, -1, 2)).permute(0, 3, 1, 2)
            yr_hat = stft.istft(norm.inorm(out, (Xr, XrMM)), n)
            loss, _, _ = io_ref.pit_neg_si_sdr(yr_hat, yr)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(leaves, 5.0)
            opt.step()
        return step
    p = ref.init_params(num_layers=8)
    leaves, seen = [], {}
    for k, v in p.items():
        if id(v) not in seen:
            seen[id(v)] = v.clone().requires_grad_(True)
            leaves.append(seen[id(v)])
        p[k] = seen[id(v)]
    opt = torch.optim.Adam(leaves, lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _, _ = io_ref.train_forward(x, yr, p, 8)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(leaves, 5.0)
        opt.step()
    return step


def cpu_baseline(seconds_budget=20.0):
    """the reference's CPU training step timed on THIS host: batch 2, fp32, all cores, and at the reference's own
    thread setting (models/utils/base_cli.py:7 pins OMP/MKL to 2 threads).  A reported baseline only; bounded sample."""
    use_ref = Path("/root/reference/models/arch/SpatialNet.py").exists()
    threads = torch.get_num_threads()
    B = 2
    step = _cpu_train_step_fn(B, use_ref)
    step()  # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 1 or (time.perf_counter() - t0 < seconds_budget and n < 4):
        step()
        n += 1
    dt = time.perf_counter() - t0
    torch.set_num_threads(2)
    step1 = _cpu_train_step_fn(1, use_ref)
    t1 = time.perf_counter()
    step1()
    dt2 = time.perf_counter() - t1
    torch.set_num_threads(threads)
    what = "the reference's own SpatialNet / STFT / Norm modules (/root/reference) + restated uPIT loss" if use_ref else \
           "the oracle (CPU restatement of the reference's training step; /root/reference is not on this box)"
    return {"value": B * n / dt, "unit": "utterances/s", "cores": threads, "kind": "reference" if use_ref else "port",
            "value_2_threads": 1.0 / dt2,
            "sample": f"{n} fp32 training steps (STFT..clip+Adam) of {what} at batch {B}, 4-s 6-ch utterances, torch CPU {threads} threads "
                      f"(os.cpu_count() = {os.cpu_count()}); value_2_threads: one step at batch 1 with 2 threads (base_cli.py:7 setting)"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU on this node"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--batch", str(args.batch)] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    # headline batch 32 (SURVEY.md §8(d) names B/GPU in {2, 8, 32}); the line also carries a short sweep over the other two
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            self_launch(args)
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl")  # RCCL on ROCm

    from models.arch.SpatialNet import SpatialNet
    from nbss_amd._lib import NBSS_BF16, hip
    from nbss_amd.engine import SpatialNetEngine, TrainStep

    lib = hip()
    torch.manual_seed(2)  # seed_everything: 2 (configs/SpatialNet.yaml:1); identical init on every rank
    net = SpatialNet(dim_input=12, dim_output=4, num_layers=8, encoder_kernel_size=5, dim_hidden=96, dim_ffn=192, num_heads=4, dropout=(0, 0, 0),
                     kernel_size=(5, 3), conv_groups=(8, 8), norms=("LN", "LN", "GN", "LN", "LN", "LN"), dim_squeeze=8, num_freqs=129, full_share=0)
    eng = SpatialNetEngine(lib, dev, dtype=NBSS_BF16, **net.hp)
    eng.load_params({k: v for k, v in net.named_parameters(remove_duplicate=False)})
    ts = TrainStep(eng, n_fft=256, ref_channel=0, lr=1e-3, clip=5.0)
    B = args.batch
    x, yr = synth_batch(B, 6, 2, 32000, 1234 + rank, dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # ---- warm-up (untimed); the first two steps are fully profiled to find the dominant kernel -------
    nk = lib.nbss_profile_kernels()
    losses, prof, nprof = [], None, 0
    for i in range(max(args.warmup, 2)):
        if i == 1:  # step 0 pays the one-time code-object loads: profile from step 1 on
            torch.cuda.synchronize()
            lib.nbss_profile_enable((1 << nk) - 1)
        losses.append(ts.step(x, yr))
        if i >= 1:
            nprof += 1
    torch.cuda.synchronize()
    prof = profile_read(lib)
    lib.nbss_profile_enable(0)
    cand = {k: v for k, v in prof.items() if algorithmic_bytes(k, B) and v[1] > 0}
    dominant = max(cand, key=lambda k: cand[k][0]) if cand else None
    dom_id = [i for i in range(nk) if lib.nbss_profile_name(i).decode() == dominant][0] if dominant else None

    # ---- timed region: EXACTLY --steps steps, events only around the dominant kernel ------------------
    if dom_id is not None:
        lib.nbss_profile_enable(1 << dom_id)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = ts.step(x, yr)
    sync()
    dt = time.perf_counter() - t0
    lib.nbss_profile_enable(0)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    timed_prof = profile_read(lib)
    final_loss = float(loss)

    if rank == 0:
        roof = None
        if dominant and timed_prof[dominant][1] > 0:
            ms, cnt = timed_prof[dominant]
            per_launch = algorithmic_bytes(dominant, B)
            ach = per_launch / (ms / cnt * 1e-3)
            total_gpu_ms = sum(v[0] for v in prof.values())
            # HBM traffic and MFMA utilisation of the dominant kernel come from separate rocprofv3 --pmc passes of THIS kernel build
            # (tools/round_artefacts.sh writes them with the commit they were measured on); a bench run cannot collect counters itself
            traffic = traffic_commit = mfma_util = None
            tfile = ROOT / "profiles" / "pmc_traffic.json"
            if tfile.exists():
                tj = json.loads(tfile.read_text())
                if tj.get("batch") == B and dominant in tj.get("kernels", {}):
                    traffic, traffic_commit = tj["kernels"][dominant]["hbm_bytes"], tj.get("commit")
            mfile = ROOT / "profiles" / "pmc_mfma.json"
            if mfile.exists():
                mj = json.loads(mfile.read_text())
                if dominant in mj.get("kernels", {}):
                    mfma_util = mj["kernels"][dominant].get("mfma_busy_frac")
            roof = {"bound": "hbm", "kernel": dominant, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                    "traffic": traffic, "traffic_commit": traffic_commit, "mfma_util": mfma_util, "avg_launch_us": ms / cnt * 1e3, "launches": cnt, "algorithmic_bytes_per_launch": per_launch,
                    "share_of_gpu_time": prof[dominant][0] / total_gpu_ms if total_gpu_ms > 0 else None,
                    "step_algorithmic": {"bytes_per_utt": 204 * S_BYTES_BF16, "achieved_GBps": world * B * args.steps / dt * 204 * S_BYTES_BF16 / 1e9 / world,
                                         "frac_of_hbm_per_gpu": (B * args.steps / dt) * 204 * S_BYTES_BF16 / HBM_PEAK}}
        base = None
        sweep = None
        if world == 1 and not args.no_cpu_baseline:
            # utterances/s at the other per-GPU batches of SURVEY.md §8(d) (short runs: 1 warm-up + 3 timed steps each)
            sweep = {str(B): round(B * args.steps / dt, 1)}
            for b2 in (2, 8, 32):
                if b2 == B:
                    continue
                x2, y2 = synth_batch(b2, 6, 2, 32000, 77, dev)
                ts.step(x2, y2)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    ts.step(x2, y2)
                torch.cuda.synchronize()
                sweep[str(b2)] = round(b2 * 3 / (time.perf_counter() - t1), 1)
            base = cpu_baseline()
        line = {
            "metric": "utterances/sec (4 s, 6ch, 129 freqs) SpatialNet bf16 train at 1/2/4/8 MI355X",
            "value": world * B * args.steps / dt, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "SpatialNet-small 6ch->2spk, 4-s 8-kHz utterances (32000 samples), n_fft 256/hop 128 (F=129, T=251), 8 layers, "
                                   f"full train step (STFT..Adam), bf16 stream + fp32 master/stats, {B} utterances per GPU per step", "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "final_loss": final_loss},
            "roofline": roof, "cpu_baseline": base, "utt_per_s_by_batch": sweep,
            "kernel_ms_per_step": {k: round(v[0] / max(nprof, 1), 4) for k, v in prof.items() if v[1] > 0},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
