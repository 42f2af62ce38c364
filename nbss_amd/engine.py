"""Host-side engine: owns the flat parameter / gradient / optimizer buffers, the packed MFMA weight
fragments, the saved activations and the workspace for one (B, F, T) geometry, and sequences the C-ABI
calls of a full training step (SharedTrainer.py:104-149 + general_steps.py:243-271 semantics):

    stft+norm -> SpatialNet fwd -> inorm+istft -> uPIT neg-SI-SDR (+grad) -> istft adjoint
    -> SpatialNet bwd -> [RCCL all-reduce of the flat fp32 gradient] -> clip + Adam -> re-pack weights

torch provides device memory, streams and torch.distributed only; every arithmetic step is a HIP kernel
behind include/nbss_hip.h.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import NBSS_BF16, NBSS_F32, Cfg, Lib, NbssError, make_cfg
from .params import param_table


class SpatialNetEngine:
    def __init__(self, lib: Lib, device, *, dim_input: int, dim_output: int, num_freqs: int, num_layers: int = 8, dim_hidden: int = 96,
                 dim_ffn: int = 192, dim_squeeze: int = 8, num_heads: int = 4, encoder_kernel_size: int = 5, kernel_size=(5, 3),
                 conv_groups=(8, 8), full_share: int = 0, dtype: int = NBSS_BF16):
        self.lib, self.device = lib, torch.device(device)
        self.kw = dict(C_in=dim_input, C_out=dim_output, H=dim_hidden, FFN=dim_ffn, SQ=dim_squeeze, L=num_layers, heads=num_heads,
                       enc_ks=encoder_kernel_size, f_ks=kernel_size[0], t_ks=kernel_size[1], f_groups=conv_groups[0], t_groups=conv_groups[1],
                       full_share=full_share)
        self.num_freqs = num_freqs
        self.dtype = dtype
        cfg0 = self.cfg_for(1, 16)
        n = lib.nbss_param_count(C.byref(cfg0))
        if n <= 0:
            raise NbssError("this SpatialNet configuration has no HIP kernels in this build: SpatialNet-small (96 / 192 / squeeze 8, training + inference) "
                            "and SpatialNet-large (192 / 384 / squeeze 16, inference + the generic backward), 4 heads, conv groups (8, 8), kernel sizes (5, 3)")
        self.table = param_table(lib, cfg0)
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.packed: Dict[int, Tensor] = {}
        self._packed_version = -1
        self.version = 0  # bump whenever params change
        self._geom: Optional[Tuple[int, int, int]] = None
        self.acts = self.ws = None
        self._slots: Dict[bool, Optional[tuple]] = {}  # train? -> (geometry key, workspace, saved activations)

    # ---- configuration / buffers -------------------------------------------------------------
    def cfg_for(self, B: int, T: int, dtype: Optional[int] = None) -> Cfg:
        return make_cfg(B, self.num_freqs, T, dtype=self.dtype if dtype is None else dtype, **self.kw)

    def stream_dtype(self, dtype: Optional[int] = None) -> torch.dtype:
        return torch.bfloat16 if (self.dtype if dtype is None else dtype) == NBSS_BF16 else torch.float32

    def load_params(self, p: Dict[str, Tensor]) -> None:
        self.params.copy_(ops.flatten_params(self.lib, self.cfg_for(1, 16), p, self.device))
        self.version += 1

    def param_views(self, flat: Tensor) -> Dict[str, Tensor]:
        return {name: flat[off:off + _numel(shape)].view(shape) for name, (off, shape) in self.table.items()}

    def packed_for(self, dtype: int) -> Tensor:
        if self._packed_version != self.version:
            self.packed.clear()
            self._packed_version = self.version
        if dtype not in self.packed:
            self.packed[dtype] = ops.pack_params(self.lib, self.cfg_for(1, 16, dtype), self.params)
        return self.packed[dtype]

    def ensure_geometry(self, B: int, T: int, train: bool, dtype: int) -> Cfg:
        cfg = self.cfg_for(B, T, dtype)
        if train and T > 256:
            raise NbssError(f"T={T}: sequences beyond 256 frames are forward-only (validate / test / predict); training keeps one whole "
                            "sequence per workgroup in LDS — cut training segments to <= 256 frames (the reference trains on 4 s = 251)")
        key = (B, T, dtype, train)
        if self._geom != key:
            # one buffer set per mode (training / inference), kept across switches: fit() alternates between a training epoch and a validation pass,
            # and reallocating GBs of workspace at every switch was pure overhead (training: the backward walk's per-sub-block workspace copies +
            # the saved activations; inference: one workspace + the two ping-pong stream buffers behind it)
            slot = self._slots.get(train)
            if slot is None or slot[0] != key:
                esz = 2 if dtype == NBSS_BF16 else 4
                infer = self.lib.nbss_workspace_bytes(C.byref(cfg)) + 2 * ((B * cfg.F * T * cfg.H * esz + 255) // 256 * 256)
                self._slots[train] = None  # (release the old set of this mode before allocating the new one)
                ws = ops.scratch(self.lib.nbss_train_ws_bytes(C.byref(cfg)) if train else infer, self.device)
                acts = ops.scratch(self.lib.nbss_acts_bytes(C.byref(cfg)), self.device) if train else None
                slot = self._slots[train] = (key, ws, acts)
            _, self.ws, self.acts = slot
            self._geom = key
        return cfg

    # ---- network ---------------------------------------------------------------------------------
    def forward(self, xin: Tensor, train: bool, dtype: Optional[int] = None) -> Tensor:
        """xin [B,F,T,C_in] (stream dtype) -> out [B,F,T,C_out] fp32.  train=True keeps the block inputs."""
        dtype = self.dtype if dtype is None else dtype
        B, F, T, _ = xin.shape
        cfg = self.ensure_geometry(B, T, train, dtype)
        out = torch.empty(B, F, T, cfg.C_out, dtype=torch.float32, device=xin.device)
        lib = self.lib
        lib.call("nbss_spatialnet_fwd", C.byref(cfg), ops._ptr(lib, self.params), ops._ptr(lib, self.packed_for(dtype)),
                 ops._ptr(lib, xin, self.stream_dtype(dtype)), ops._ptr(lib, self.acts), ops._ptr(lib, self.ws), ops._ptr(lib, out),
                 ops._stream(lib, xin))
        return out

    def backward(self, xin: Tensor, dout: Tensor, dtype: Optional[int] = None, on_bucket=None) -> None:
        """accumulates parameter gradients of the LAST train-mode forward into self.grads.

        on_bucket(lo, hi): when given, backward is walked one layer at a time (nbss_spatialnet_bwd_range) and the callback is
        invoked as soon as grads[lo:hi] is final — the decoder with the last layer, the encoder and the shared LinearGroup with
        layer 0 — so that a data-parallel caller can start reducing that slice while the remaining layers still run."""
        dtype = self.dtype if dtype is None else dtype
        B, F, T, _ = xin.shape
        if self._geom != (B, T, dtype, True):
            raise NbssError("backward() needs a preceding forward(train=True) with the same geometry")
        cfg = self.cfg_for(B, T, dtype)
        lib = self.lib
        args = (C.byref(cfg), ops._ptr(lib, self.params), ops._ptr(lib, self.grads), ops._ptr(lib, self.packed_for(dtype)),
                ops._ptr(lib, xin, self.stream_dtype(dtype)), ops._ptr(lib, self.acts), ops._ptr(lib, dout, torch.float32), ops._ptr(lib, self.ws))
        if on_bucket is None:
            lib.call("nbss_spatialnet_bwd", *args, ops._stream(lib, xin))
            return
        L = cfg.L
        for l in range(L - 1, -1, -1):
            lib.call("nbss_spatialnet_bwd_range", *args, l + 1, l, ops._stream(lib, xin))
            for lo, hi in self.grad_buckets()[l]:
                on_bucket(lo, hi)

    def grad_buckets(self):
        """per layer: the [lo, hi) ranges of the flat gradient buffer that are final once that layer's backward has run
        (reverse-order buckets of SURVEY.md §8(e): last layer + decoder first, layer 0 + encoder + shared LinearGroup last)"""
        if getattr(self, "_buckets", None) is None:
            L = self.kw["L"]
            owner = {}  # a tensor shared by several layers (full_share) is final only after the LOWEST layer that uses it
            for name, (off, shape) in self.table.items():
                if name.startswith("layers."):
                    key = int(name.split(".")[1])
                elif name.startswith("decoder"):
                    key = L - 1
                else:  # encoder
                    key = 0
                span = (off, off + _numel(shape))
                owner[span] = min(owner.get(span, key), key)
            spans = {}
            for span, key in owner.items():
                spans.setdefault(key, []).append(span)
            buckets = []
            covered = 0
            for l in range(L):
                merged = []
                for lo, hi in sorted(spans.get(l, [])):
                    if merged and merged[-1][1] == lo:
                        merged[-1] = (merged[-1][0], hi)
                    else:
                        merged.append((lo, hi))
                covered += sum(hi - lo for lo, hi in merged)
                buckets.append(merged)
            assert covered == self.params.numel(), "gradient buckets must tile the flat buffer"
            self._buckets = buckets
        return self._buckets


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= s
    return n


class TrainStep:
    """One full SpatialNet training step on the HIP path (the unit bench.py times)."""

    def __init__(self, engine: SpatialNetEngine, *, n_fft: int = 256, ref_channel: int = 0, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, clip: float = 5.0, process_group=None, bucketed: bool = True, force_collectives: bool = False,
                 decoupled_weight_decay: bool = False, window: int = 0, graph: Optional[bool] = None):
        self.e = engine
        self.lib = engine.lib
        self.n_fft, self.ref = n_fft, ref_channel
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip
        self.tables = ops.stft_tables(self.lib, n_fft, window, engine.device)  # window: 0 hann, 1 sqrt-hann (models/io/stft.py)
        self.decoupled_wd = decoupled_weight_decay
        self.m = torch.zeros_like(engine.params)
        self.v = torch.zeros_like(engine.params)
        self.scratch = torch.zeros(512, dtype=torch.float32, device=engine.device)
        self.step_count = 0
        self.pg = process_group
        self.bucketed = bucketed  # world > 1: per-layer gradient buckets reduced while backward is still running
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # collectives run when world > 1; tests force them on a 1-rank RCCL group to exercise the stream ordering on hardware
        self.collectives = self.world > 1 or force_collectives
        self.comm_wait_ms = None  # set to 0.0 by a caller that wants the host-visible time of the bucket waits accumulated (bench.py)
        # HIP-graph replay of the step (graph_step below): opt-in.  Measured on MI355X / ROCm 7.2 (profiles/README.md round 4): the eager
        # two-stream step is FASTER at every batch — batch 2: 5.77 ms eager, 6.09 ms replayed in order, 13.6 ms replayed with the walks'
        # forks and joins captured (~45 us per kernel node of a branched graph); batch 8: 14.6 vs 15.2 ms — so None (the default) means off;
        # NBSS_GRAPH=1 or graph=True turns it on (correct: tests/test_graph_step.py)
        self.graph = graph
        self._graphs = {}

    # ---- replica consistency (what Lightning's DDP does at fit start: broadcast of the module state from rank 0) --------
    def sync_replicas(self, src: int = 0) -> None:
        """rank `src`'s parameters, Adam moments and step count become every rank's (identical seeds already make them equal; this makes
        it a guarantee, e.g. after a resume from a checkpoint only rank 0 read)"""
        if not self.collectives:
            return
        e = self.e
        for t in (e.params, self.m, self.v):
            torch.distributed.broadcast(t, src=src, group=self.pg)
        sc = torch.tensor([float(self.step_count), float(self.lr)], dtype=torch.float64, device=e.device)
        torch.distributed.broadcast(sc, src=src, group=self.pg)
        self.step_count, self.lr = int(sc[0].item()), float(sc[1].item())
        e.version += 1
        e.packed_for(e.dtype)

    def replica_checksum(self) -> Tensor:
        """[sum(params), sum(|params|), sum(exp_avg), sum(exp_avg_sq)] in fp64 on the device"""
        p = self.e.params.double()
        return torch.stack([p.sum(), p.abs().sum(), self.m.double().sum(), self.v.double().sum()])

    def check_replicas(self) -> float:
        """every rank must hold bitwise the same parameters and optimizer state (each applies the same reduced gradient with the same
        kernel): all-reduces the min and the max of a checksum and raises when they differ; returns the spread (0.0 when consistent)"""
        if not self.collectives:
            return 0.0
        c = self.replica_checksum()
        lo, hi = c.clone(), c.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN, group=self.pg)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX, group=self.pg)
        spread = float((hi - lo).abs().max())
        if spread != 0.0:
            raise RuntimeError(f"data-parallel replicas diverged: checksum spread {spread:g} (min {lo.tolist()}, max {hi.tolist()})")
        return spread

    def forward_loss(self, x: Tensor, yr: Tensor, need_grad: bool = True):
        """x [B,C,N] fp32 mixture, yr [B,S,N] fp32 targets -> (loss [1], yr_hat [B,S,N], dout or None, xin, xrmm)"""
        e, lib = self.e, self.lib
        N = x.shape[-1]
        xin, xrmm = ops.stft_norm_fwd(lib, self.n_fft, e.dtype, self.tables, x, self.ref)
        out = e.forward(xin, train=need_grad)
        yr_hat = ops.inorm_istft_fwd(lib, self.n_fft, self.tables, out, xrmm, N)
        loss, perm, dyh = ops.pit_neg_sisdr(lib, yr_hat, yr, need_grad=need_grad)
        dout = ops.inorm_istft_bwd(lib, self.n_fft, self.tables, dyh, xrmm) if need_grad else None
        return loss, yr_hat, dout, xin, perm

    def step(self, x: Tensor, yr: Tensor) -> Tensor:
        """forward + backward + (all-reduce) + clip + Adam + re-pack; returns the loss (device tensor [1])"""
        if self._use_graph(x):
            return self.graph_step(x, yr)
        e = self.e
        loss, _, dout, xin, _ = self.forward_loss(x, yr, need_grad=True)
        self.backward_and_update(xin, dout)
        return loss

    # ---- HIP-graph replay ---------------------------------------------------------------------------------------------------------------
    # The step is a fixed launch sequence per input shape: STFT / network walk (its two library-owned streams fork and join inside the walk
    # calls) / loss / walk back, then clip + Adam + re-pack.  It is captured as TWO graphs around the gradient exchange — A: forward + loss +
    # backward into engine.grads; B: the optimizer — so that with world > 1 the RCCL all-reduce (one collective, outside any capture) sits
    # between two replays.  Per-step scalars (learning rate, Adam bias corrections) travel through a 3-float device buffer, everything else
    # is baked in; a change of shape, hyper-parameter or parameter storage captures afresh.
    def _use_graph(self, x: Tensor) -> bool:
        if not x.is_cuda or self.comm_wait_ms is not None:
            return False
        if self.graph is None:
            return os.environ.get("NBSS_GRAPH") == "1"  # A/B knob of the tools
        return bool(self.graph)

    def graph_step(self, x: Tensor, yr: Tensor) -> Tensor:
        e = self.e
        key = (tuple(x.shape), tuple(yr.shape), e.dtype, self.betas, self.eps, self.wd, self.clip, self.decoupled_wd, self.world, e.params.data_ptr())
        g = self._graphs.get(key)
        if g is None or g["state"] == 0:
            # first call with this key: one eager step (lazy one-time work — workspace allocation, kernel attributes, the library's streams —
            # must not happen under capture); second call: capture, then replay
            if g is None:
                self._graphs[key] = {"state": 0}
                e = self.e
                loss, _, dout, xin, _ = self.forward_loss(x, yr, need_grad=True)
                self.backward_and_update(xin, dout)
                return loss
            g = self._capture(key, x, yr)
        if e.packed.get(e.dtype) is not g["packed"] or e._packed_version != e.version:
            # something else re-packed in between (an eager step of another shape, load_params): bring the graph's own buffer up to date
            ops.pack_params(self.lib, e.cfg_for(1, 16, e.dtype), e.params, out=g["packed"])
            e.packed.clear()
            e.packed[e.dtype] = g["packed"]
            e._packed_version = e.version
        g["x"].copy_(x, non_blocking=True)
        g["yr"].copy_(yr, non_blocking=True)
        g["a"].replay()
        if self.collectives:
            torch.distributed.all_reduce(e.grads, group=self.pg)
        self.step_count += 1
        # per-step scalars: a small ring of pinned host buffers, each guarded by an event recorded behind its copy — the host may run several steps
        # ahead of the device (no per-step sync), and a single buffer would be rewritten under a pending DMA
        i = self.step_count % len(g["hyper_host"])
        hh, ev = g["hyper_host"][i], g["hyper_ev"][i]
        ev.synchronize()  # (no-op for an event that was never recorded)
        self.lib.call("nbss_adam_hyper", int(self.step_count), float(self.lr), float(self.betas[0]), float(self.betas[1]), hh.data_ptr())
        g["hyper"].copy_(hh, non_blocking=True)
        ev.record()
        g["b"].replay()
        e.version += 1
        e._packed_version = e.version  # the replayed re-pack refreshed the fragments in place
        return g["loss"].clone()

    def _capture(self, key, x: Tensor, yr: Tensor) -> dict:
        e = self.e
        g = {"state": 1, "x": torch.empty_like(x), "yr": torch.empty_like(yr), "hyper": torch.zeros(4, dtype=torch.float32, device=x.device),
             "hyper_host": [torch.zeros(4, dtype=torch.float32).pin_memory() for _ in range(4)], "hyper_ev": [torch.cuda.Event() for _ in range(4)]}
        g["x"].copy_(x)
        g["yr"].copy_(yr)
        g["packed"] = e.packed_for(e.dtype)
        # the training workspace / saved activations of THIS shape, allocated eagerly (another shape may have taken the engine's slot since the
        # eager first step) and held by the graph: its kernels have these addresses baked in, and a later change of shape drops the engine's
        # reference (ensure_geometry) — without the graph's own the replay would write into memory returned to the allocator
        xin0, _ = ops.stft_norm_fwd(self.lib, self.n_fft, e.dtype, self.tables, g["x"], self.ref)
        e.ensure_geometry(xin0.shape[0], xin0.shape[2], True, e.dtype)
        g["ws"], g["acts"] = e.ws, e.acts
        del xin0
        torch.cuda.synchronize()
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            loss, _, dout, xin, _ = self.forward_loss(g["x"], g["yr"], need_grad=True)
            e.backward(xin, dout)
        with torch.cuda.graph(gb, pool=ga.pool()):
            ops.clip_adam_step_dev(self.lib, e.params, e.grads, self.m, self.v, self.scratch, g["hyper"], betas=self.betas, eps=self.eps, weight_decay=self.wd,
                                   max_norm=self.clip, grad_scale=1.0 / self.world, zero_grad=True, decoupled_weight_decay=self.decoupled_wd)
            # re-pack IN PLACE: graph A reads this very buffer (a fresh one per step, as the eager path allocates, would leave the replayed
            # forward on the fragments of the capture step)
            ops.pack_params(self.lib, e.cfg_for(1, 16, e.dtype), e.params, out=g["packed"])
        assert e.ws is g["ws"] and e.acts is g["acts"]  # (nothing re-allocated under capture)
        e.version += 1
        e._packed_version = e.version
        g.update(a=ga, b=gb, loss=loss)
        self._graphs[key] = g
        return g

    def backward_and_update(self, xin: Tensor, dout: Tensor) -> None:
        """network backward + gradient exchange + clip/Adam/re-pack"""
        e = self.e
        if self.collectives and self.bucketed:
            # data parallel, overlapped: one asynchronous all-reduce (SUM) per layer bucket, issued as soon as that layer's
            # backward has been enqueued; RCCL runs them on its own stream behind the kernels already queued here
            handles = []
            e.backward(xin, dout, on_bucket=lambda lo, hi: handles.append(
                torch.distributed.all_reduce(e.grads[lo:hi], group=self.pg, async_op=True)))
            if self.comm_wait_ms is None:
                for h in handles:
                    h.wait()
            else:  # how long the step's own stream has to wait for the exchange AFTER backward is enqueued: events around the waits
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for h in handles:
                    h.wait()
                e1.record()
                self._comm_events = getattr(self, "_comm_events", []) + [(e0, e1)]
            self.apply_gradients(reduced=True)
        else:
            e.backward(xin, dout)
            self.apply_gradients()

    def comm_wait_read(self) -> float:
        """ms the compute stream spent in the bucket waits since the last call (needs comm_wait_ms = 0.0 before the steps)"""
        ev = getattr(self, "_comm_events", [])
        self._comm_events = []
        if not ev:
            return 0.0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in ev))

    def apply_gradients(self, reduced: bool = False) -> None:
        """[all-reduce] + clip + Adam + re-pack on whatever is in engine.grads (reduced=True: the buckets were summed already)"""
        e = self.e
        if self.collectives and not reduced:
            # data parallel: ONE all-reduce (SUM) of the flat fp32 gradient over RCCL; the mean is folded into the clip kernel
            torch.distributed.all_reduce(e.grads, group=self.pg)
        self.step_count += 1
        ops.clip_adam_step(self.lib, e.params, e.grads, self.m, self.v, self.scratch, self.step_count, lr=self.lr, betas=self.betas, eps=self.eps,
                           weight_decay=self.wd, max_norm=self.clip, grad_scale=1.0 / self.world, zero_grad=True, decoupled_weight_decay=self.decoupled_wd)
        e.version += 1
        e.packed_for(e.dtype)  # re-pack inside the step: the next forward needs fresh fragments
