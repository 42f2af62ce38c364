"""SharedTrainer — drop-in for the reference's SharedTrainer.py surface on MI355X:
  * `TrainModule(arch, channels, ref_channel, stft, norm, loss, optimizer, lr_scheduler, ...)` with
    `forward(x, istft=True)` / `training_step` (reference :38-63,104-149);
  * `python SharedTrainer.py fit|test|predict --config a.yaml --config b.yaml --model.arch.dim_input=12 ...`
    (reference :344-382, README.md:46-57): YAML `class_path` / `init_args` instantiation, dotted overrides.
pytorch_lightning / jsonargparse are not required (they are absent on the target image); the small CLI below parses the
same flags.  `fit` runs the fused HIP training step (nbss_amd.engine.TrainStep): one process per GPU, gradients
all-reduced over RCCL when launched under torchrun.
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import time
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
import yaml
from torch import Tensor

from models.io.loss import Loss, neg_si_sdr
from models.io.norm import Norm
from models.io.stft import STFT


class _FusedIO(torch.autograd.Function):
    """inorm + iSTFT on the network output (fp32), differentiable w.r.t. `out`"""

    @staticmethod
    def forward(ctx, out, xrmm, tables, n_fft, N):
        from nbss_amd import ops
        from nbss_amd._lib import hip
        ctx.args = (xrmm, tables, n_fft)
        return ops.inorm_istft_fwd(hip(), n_fft, tables, out.float().contiguous(), xrmm, N)

    @staticmethod
    def backward(ctx, dy):
        from nbss_amd import ops
        from nbss_amd._lib import hip
        xrmm, tables, n_fft = ctx.args
        return ops.inorm_istft_bwd(hip(), n_fft, tables, dy.contiguous(), xrmm), None, None, None, None


class TrainModule(nn.Module):
    name: str
    import_path: str = "SharedTrainer.TrainModule"

    def __init__(self, arch: nn.Module, channels: List[int], ref_channel: int, stft: STFT = None, norm: Norm = None, loss: Loss = None,
                 optimizer: Tuple[str, Dict[str, Any]] = ("Adam", {"lr": 0.001}), lr_scheduler: Optional[Tuple[str, Dict[str, Any]]] = None,
                 metrics: List[str] = ("SDR", "SI_SDR"), mchunk=None, val_metric: str = "loss", write_examples: int = 200, ensemble=None,
                 compile: bool = False, exp_name: str = "exp", reset: Optional[List[str]] = None):
        super().__init__()
        self.arch = arch  # `compile` is accepted and ignored: the arch already is a fused native kernel graph
        self.channels, self.ref_channel = list(channels), ref_channel
        self.stft = stft if stft is not None else STFT(n_fft=256, n_hop=128, win_len=256)
        self.norm = norm if norm is not None else Norm(mode="utterance")
        self.loss = loss if loss is not None else Loss(loss_func=neg_si_sdr, pit=True)
        self.optimizer, self.lr_scheduler = optimizer, lr_scheduler
        self.metrics, self.val_metric, self.exp_name, self.reset = list(metrics), val_metric, exp_name, reset
        self.name = type(arch).__name__
        self.precision = "32"

    def _fusable(self) -> bool:
        return self.norm.mode == "frequency" and self.norm.online and self.loss.mask is None

    def _fused_path(self, x: Tensor) -> bool:
        from models.arch.SpatialNet import SpatialNet
        return isinstance(self.arch, SpatialNet) and x.is_cuda and self._fusable() and self.stft.hip_ok

    def forward(self, x: Tensor, istft: bool = True):
        """x [B,C,N] -> (yr_hat [B,Spk,N], loss_paras)   (reference :104-132).  SpatialNet on a HIP device with the shipped I/O
        configuration takes the fused kernels (STFT+norm, network, inorm+iSTFT); every other combination — the narrow-band archs,
        other Norm modes, host tensors — walks the reference's module sequence (stft -> norm -> arch -> inorm -> istft)."""
        if not self._fused_path(x):
            return self._forward_modules(x, istft)
        from nbss_amd import ops
        from nbss_amd._lib import NBSS_BF16, NBSS_F32, hip
        xs = x[:, self.channels].float().contiguous()
        N = xs.shape[-1]
        tables = self.stft._tables(xs.device)
        bf16 = self.precision in ("bf16-mixed", "bf16")
        X, xrmm = ops.stft_norm_fwd(hip(), self.stft.n_fft, NBSS_BF16 if bf16 else NBSS_F32, tables, xs, self.channels.index(self.ref_channel))
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            out = self.arch(X)
        if not istft:
            B, F, T, S2 = out.shape
            return out.float() * xrmm[..., None], {"XrMM": xrmm}
        return _FusedIO.apply(out, xrmm, tables, self.stft.n_fft, N), {"XrMM": xrmm}

    def _forward_modules(self, x: Tensor, istft: bool = True):
        X, length = self.stft.stft(x[:, self.channels])  # [B,C,F,T] complex
        B, C, F, T = X.shape
        X, (Xr, XrMM) = self.norm.norm(X, ref_channel=self.channels.index(self.ref_channel))
        feats = torch.view_as_real(X.permute(0, 2, 3, 1).contiguous()).reshape(B, F, T, 2 * C)
        bf16 = self.precision in ("bf16-mixed", "bf16") and x.is_cuda
        with torch.autocast(x.device.type, dtype=torch.bfloat16, enabled=bf16):
            out = self.arch(feats)
        if not torch.is_complex(out):
            out = torch.view_as_complex(out.float().reshape(B, F, T, -1, 2).contiguous())
        out = out.permute(0, 3, 1, 2)  # [B,Spk,F,T]
        Yr_hat, loss_paras = self.loss.to_CC(out=out, Xr=Xr, XrMM=XrMM, stft=self.stft)
        if self.loss.mask is None:
            Yr_hat = self.norm.inorm(out, (Xr, XrMM))
        return (self.stft.istft(Yr_hat, length) if istft else torch.view_as_real(Yr_hat)), loss_paras

    @torch.no_grad()
    def forward_streaming(self, x: Tensor, chunk: int = 8, use_graph: Optional[bool] = None, native: Optional[bool] = None):
        """causal chunked inference of an OnlineSpatialNet: x [B,C,N] -> (yr_hat [B,Spk,N], stats).  STFT and the (online) normalisation
        are frame-local, so the network is fed `chunk` frames at a time through OnlineStreamer (fixed-shape state; one HIP graph replay
        per chunk on a HIP device) and the result equals forward() on the whole signal (reference OnlineSpatialNet.py:333-354)."""
        from models.arch.OnlineSpatialNet import OnlineStreamer
        assert self.norm.online or self.norm.mode in ("none", None), "streaming needs a causal (online) normalisation"
        X, length = self.stft.stft(x[:, self.channels])
        B, C, F, T = X.shape
        X, (Xr, XrMM) = self.norm.norm(X, ref_channel=self.channels.index(self.ref_channel))
        feats = torch.view_as_real(X.permute(0, 2, 3, 1).contiguous()).reshape(B, F, T, 2 * C)
        Tp = (T + chunk - 1) // chunk * chunk
        if Tp != T:
            feats = torch.nn.functional.pad(feats, (0, 0, 0, Tp - T))
        # the native streamer keeps packed COPIES of the weights: the key carries a weights version (every in-place update — optimizer step,
        # load_state_dict, load_checkpoint — bumps a parameter's _version), so a cached streamer never serves stale weights
        wver = sum(int(p._version) for p in self.arch.parameters()) + sum(int(b._version) for b in self.arch.buffers())
        key = (B, chunk, str(x.device), use_graph, native, wver, tuple(id(p) for p in self.arch.parameters()))
        if getattr(self, "_streamer_key", None) != key:
            # on a HIP device the native step (nbss_amd/online.py: HIP kernels for the causal encoder, the recurrent retention and the causal
            # T-ConvFFN + the cross-band kernels, one HIP graph per chunk) serves the geometry it is built for; everything else — other
            # attention types / widths, host tensors — takes the torch.nn step (OnlineStreamer)
            from nbss_amd.online import NativeOnlineStreamer, supported
            if x.is_cuda and supported(self.arch) is None and chunk <= 32 and native is not False:
                self._streamer = NativeOnlineStreamer(self.arch, B, chunk, device=x.device, use_graph=use_graph)
            else:
                self._streamer = OnlineStreamer(self.arch, B, chunk, device=x.device, use_graph=use_graph)
            self._streamer_key = key
        s = self._streamer
        s.reset()
        if x.is_cuda:
            torch.cuda.synchronize()
        t0 = time.time()
        out = torch.cat([s.step(feats[:, :, c:c + chunk]) for c in range(0, Tp, chunk)], 2)[:, :, :T]
        if x.is_cuda:
            torch.cuda.synchronize()
        dt = time.time() - t0
        out = torch.view_as_complex(out.float().reshape(B, F, T, -1, 2).contiguous()).permute(0, 3, 1, 2)
        Yr_hat = self.norm.inorm(out, (Xr, XrMM))
        stats = {"chunks": Tp // chunk, "graph_replays": Tp // chunk if s.graph is not None else 0, "frames_per_s": B * Tp / max(dt, 1e-9),
                 "native": type(s).__name__ == "NativeOnlineStreamer"}
        return self.stft.istft(Yr_hat, length), stats

    def training_step(self, batch, batch_idx=0):
        x, ys, paras = batch
        yr = ys[:, :, self.ref_channel, :]
        yr_hat, _ = self.forward(x)
        loss, perms, _ = self.loss(yr_hat=yr_hat, yr=yr, reorder=False, reduce_batch=True)
        return loss

    def configure_optimizers(self):
        name, kw = self.optimizer
        opt = getattr(torch.optim, name)(self.parameters(), **kw)
        if self.lr_scheduler:
            sname, skw = self.lr_scheduler
            return opt, getattr(torch.optim.lr_scheduler, sname)(opt, **skw)
        return opt, None


# ---------------------------------------------------------------------------------------------------
# minimal LightningCLI-compatible command line
def _deep_merge(a: dict, b: dict) -> dict:
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _deep_merge(a[k], v)
        else:
            a[k] = v
    return a


def _set_dotted(cfg: dict, key: str, value: Any) -> None:
    parts = key.split(".")
    d = cfg
    for i, p in enumerate(parts[:-1]):
        nxt = d.get(p)
        if isinstance(nxt, dict) and "class_path" in nxt and parts[i + 1] not in ("class_path", "init_args"):
            nxt = nxt.setdefault("init_args", {})
            d = nxt
            continue
        if not isinstance(nxt, dict):
            nxt = d[p] = {}
        d = nxt
    d[parts[-1]] = value


def _instantiate(node: Any) -> Any:
    if isinstance(node, dict) and "class_path" in node:
        mod, _, cls = node["class_path"].rpartition(".")
        kwargs = {k: _instantiate(v) for k, v in (node.get("init_args") or {}).items()}
        return getattr(importlib.import_module(mod), cls)(**kwargs)
    if isinstance(node, dict):
        return {k: _instantiate(v) for k, v in node.items()}
    if isinstance(node, str) and node.count(".") >= 2 and node.split(".")[0] in ("models", "data_loaders"):
        mod, _, fn = node.rpartition(".")  # callable path, e.g. models.io.loss.neg_si_sdr
        try:
            return getattr(importlib.import_module(mod), fn)
        except Exception:
            return node
    return node


def parse_cli(argv: List[str]) -> Tuple[str, dict]:
    sub, cfg, i = argv[0], {}, 1
    assert sub in ("fit", "test", "predict", "validate"), sub
    while i < len(argv):
        a = argv[i]
        assert a.startswith("--"), a
        if "=" in a:
            k, v = a[2:].split("=", 1)
            i += 1
        else:
            k, v = a[2:], argv[i + 1]
            i += 2
        if k == "config":
            with open(v) as f:
                _deep_merge(cfg, yaml.safe_load(f) or {})
        else:
            _set_dotted(cfg, k, yaml.safe_load(v))
    return sub, cfg


def build_module(cfg: dict) -> TrainModule:
    m = dict(cfg["model"])
    kw = {k: _instantiate(v) for k, v in m.items()}
    for k in ("optimizer", "lr_scheduler"):
        if isinstance(kw.get(k), list):
            kw[k] = tuple(kw[k])
    return TrainModule(**kw)


def _unique_params(module: "TrainModule"):
    """(name, parameter) in torch.optim order: what `Adam(module.parameters())` indexes its state by (shared tensors once)"""
    return list(module.named_parameters())


def save_checkpoint(path: str, module: "TrainModule", ts=None, epoch: int = 0, global_step: int = 0, plateau: "Optional[_Plateau]" = None) -> None:
    """Lightning-shaped checkpoint that the reference's trainer can load: `state_dict` with the reference's keys (`arch.*` and the
    persistent `stft.window` buffer, general_steps.py:189-199), `optimizer_states[0]` as a torch.optim.Adam state_dict (per-parameter
    `exp_avg` / `exp_avg_sq` / `step` sliced out of the fused optimizer's flat buffers, in `module.parameters()` order),
    `lr_schedulers` (a full ExponentialLR — or, with `plateau`, ReduceLROnPlateau — state_dict), `epoch`, `global_step`, `pytorch-lightning_version`.  Weights, optimizer and
    scheduler are Lightning-resumable; `loops` (Lightning's fit-loop progress counters) is left empty — Lightning then restarts its
    epoch counter from `epoch`."""
    sd = {"arch." + k: v.detach().cpu().clone() for k, v in module.arch.state_dict().items()}
    sd["stft.window"] = module.stft.window.detach().cpu().clone()
    ck = {"epoch": epoch, "global_step": global_step, "pytorch-lightning_version": "2.0.0", "state_dict": sd, "loops": {}, "callbacks": {},
          "hyper_parameters": {}}
    if ts is not None:
        eng = ts.e
        state = {}
        for i, (name, p) in enumerate(_unique_params(module)):
            off, _ = eng.table[name.removeprefix("arch.")]
            n = p.numel()
            state[i] = {"step": torch.tensor(float(ts.step_count)), "exp_avg": ts.m[off:off + n].view_as(p).detach().cpu().clone(),
                        "exp_avg_sq": ts.v[off:off + n].view_as(p).detach().cpu().clone()}
        group = {"lr": float(ts.lr), "betas": tuple(ts.betas), "eps": ts.eps, "weight_decay": ts.wd, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(state)))}
        ck["optimizer_states"] = [{"state": state, "param_groups": [group]}]
        if plateau is not None:
            # a torch.optim.lr_scheduler.ReduceLROnPlateau.state_dict(): best / bad-epoch count / cooldown travel with the checkpoint
            ck["lr_schedulers"] = [plateau.state_dict(ts.lr)]
        else:
            # a complete torch.optim.lr_scheduler.ExponentialLR.state_dict() (what Lightning stores and restores)
            gamma = float((module.lr_scheduler or (None, {}))[1].get("gamma", 1.0)) if module.lr_scheduler and module.lr_scheduler[0] == "ExponentialLR" else 1.0
            base_lr = float(module.optimizer[1].get("lr", 1e-3))
            ck["lr_schedulers"] = [{"gamma": gamma, "base_lrs": [base_lr], "last_epoch": epoch + 1, "verbose": False, "_step_count": epoch + 2,
                                    "_get_lr_called_within_step": False, "_last_lr": [float(ts.lr)]}]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ck, path)


def load_checkpoint(path: str, module: "TrainModule", ts=None, plateau: "Optional[_Plateau]" = None) -> int:
    """weights and — when `ts` is given — the Adam moments / step / decayed lr of a checkpoint written by save_checkpoint OR by the
    reference's Lightning trainer (same layout: torch.optim state indexed in `module.parameters()` order); returns the epoch to resume
    after.  A checkpoint without optimizer state resumes with fresh moments and says so."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck.get("state_dict", ck)
    sd = {k.replace("_orig_mod.", "").removeprefix("arch."): v for k, v in sd.items() if not k.endswith("stft.window")}
    module.arch.load_state_dict(sd, strict=True)
    if ts is not None:
        opt = (ck.get("optimizer_states") or [None])[0]
        if opt and "state" in opt and len(opt["state"]) > 0:
            eng = ts.e
            names = _unique_params(module)
            if len(opt["state"]) != len(names):
                raise RuntimeError(f"{path}: optimizer state has {len(opt['state'])} entries, the model has {len(names)} parameters")
            for i, (name, p) in enumerate(names):
                st = opt["state"][i]
                off, _ = eng.table[name.removeprefix("arch.")]
                ts.m[off:off + p.numel()].copy_(st["exp_avg"].reshape(-1).to(ts.m.device))
                ts.v[off:off + p.numel()].copy_(st["exp_avg_sq"].reshape(-1).to(ts.v.device))
            ts.step_count = int(float(opt["state"][0]["step"]))
            ts.lr = float(opt["param_groups"][0]["lr"])
        else:
            print(f"[SharedTrainer] {path} has no optimizer state: Adam moments, step count and learning rate start fresh", flush=True)
    if plateau is not None and ck.get("lr_schedulers"):
        plateau.load_state_dict(ck["lr_schedulers"][0])
    return int(ck.get("epoch", -1))


def _fused_step_for(module: "TrainModule", cfg: dict, dev):
    """engine.TrainStep for a TrainModule whose configuration the fused HIP step implements; everything else raises HERE (the fused
    step hard-wires Norm('frequency', online) + uPIT neg-SI-SDR, so a different YAML must not train a different model silently)"""
    from nbss_amd._lib import NBSS_BF16, NBSS_F32
    from nbss_amd.engine import TrainStep
    from models.arch.SpatialNet import SpatialNet
    tr = cfg.get("trainer", {})
    if not isinstance(module.arch, SpatialNet):
        raise NotImplementedError(f"the fused MI355X step serves models.arch.SpatialNet.SpatialNet, not {type(module.arch).__name__}")
    if not module._fusable() or not module.loss.pit:
        raise NotImplementedError("the fused MI355X step implements norm = Norm('frequency', online=True) and loss = Loss(neg_si_sdr, pit=True) "
                                  f"(configs/SpatialNet.yaml); got norm=({module.norm.mode}, online={module.norm.online}), pit={module.loss.pit}")
    eng = module.arch._engine_for(dev)
    eng.dtype = NBSS_BF16 if module.precision in ("bf16-mixed", "bf16") else NBSS_F32
    oname, okw = module.optimizer
    if oname not in ("Adam", "AdamW"):
        raise NotImplementedError(f"optimizer {oname}: the fused kernel implements torch.optim.Adam and torch.optim.AdamW")
    wd = okw.get("weight_decay", 0.01 if oname == "AdamW" else 0.0)
    gamma = 1.0
    if module.lr_scheduler:
        sname, skw = module.lr_scheduler
        if sname == "ExponentialLR":
            gamma = float(skw.get("gamma", 1.0))
        elif sname == "ReduceLROnPlateau":  # on the monitored validation metric (general_steps.py:259-271): handled by fit()'s _Plateau
            gamma = _Plateau(**{k: v for k, v in skw.items() if k not in ("verbose", "optimizer")})  # unknown arguments raise, none is dropped
        else:
            raise NotImplementedError(f"lr_scheduler {sname}: ExponentialLR (configs/SpatialNet.yaml) and ReduceLROnPlateau are implemented")
    ts = TrainStep(eng, n_fft=module.stft.n_fft, ref_channel=module.channels.index(module.ref_channel), lr=okw.get("lr", 1e-3),
                   betas=tuple(okw.get("betas", (0.9, 0.999))), eps=okw.get("eps", 1e-8), weight_decay=wd,
                   decoupled_weight_decay=oname == "AdamW", clip=float(tr.get("gradient_clip_val") or 0.0),
                   window=0 if module.stft.win == "hann_window" else 1)
    return eng, ts, gamma


class _Plateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau's rule on a plain float lr (the fused optimizer has no torch param_groups): mode, factor,
    patience, threshold + threshold_mode (rel | abs), cooldown, min_lr and the eps rule (a reduction smaller than eps is ignored).
    state_dict() / load_state_dict() use torch's field names, so a checkpoint written here resumes under Lightning and vice versa."""

    def __init__(self, mode: str = "min", factor: float = 0.1, patience: int = 10, threshold: float = 1e-4, threshold_mode: str = "rel",
                 cooldown: int = 0, min_lr: float = 0.0, eps: float = 1e-8, **unknown):
        if unknown:
            raise NotImplementedError(f"ReduceLROnPlateau arguments {sorted(unknown)} are not implemented by the fused step")
        if mode not in ("min", "max") or threshold_mode not in ("rel", "abs"):
            raise ValueError(f"ReduceLROnPlateau: mode={mode!r}, threshold_mode={threshold_mode!r}")
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        if isinstance(min_lr, (list, tuple)):
            if len(min_lr) != 1:
                raise NotImplementedError("the fused optimizer has one parameter group: min_lr must be one value")
            min_lr = min_lr[0]
        self.mode, self.factor, self.patience, self.threshold, self.threshold_mode = mode, float(factor), int(patience), float(threshold), threshold_mode
        self.cooldown, self.min_lr, self.eps = int(cooldown), float(min_lr), float(eps)
        self.best = float("inf") if mode == "min" else -float("inf")
        self.bad, self.cool, self.last_epoch, self.last_lr = 0, 0, 0, None

    def _is_better(self, a: float) -> bool:
        if self.mode == "min":
            return a < (self.best * (1.0 - self.threshold) if self.threshold_mode == "rel" else self.best - self.threshold)
        return a > (self.best * (self.threshold + 1.0) if self.threshold_mode == "rel" else self.best + self.threshold)

    def step(self, metric: float, lr: float) -> float:
        metric = float(metric)
        self.last_epoch += 1
        if self._is_better(metric):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.cool > 0:
            self.cool, self.bad = self.cool - 1, 0  # bad epochs are ignored in cooldown
        if self.bad > self.patience:
            new_lr = max(lr * self.factor, self.min_lr)
            if lr - new_lr > self.eps:
                lr = new_lr
            self.cool, self.bad = self.cooldown, 0
        self.last_lr = lr
        return lr

    def state_dict(self, lr: float) -> dict:
        return {"factor": self.factor, "min_lrs": [self.min_lr], "default_min_lr": self.min_lr, "patience": self.patience, "cooldown": self.cooldown,
                "eps": self.eps, "last_epoch": self.last_epoch, "_last_lr": [float(lr)], "mode_worse": float("inf") if self.mode == "min" else -float("inf"),
                "mode": self.mode, "threshold": self.threshold, "threshold_mode": self.threshold_mode, "best": self.best,
                "cooldown_counter": self.cool, "num_bad_epochs": self.bad}

    def load_state_dict(self, sd: dict) -> None:
        if "num_bad_epochs" not in sd:
            raise RuntimeError("checkpoint's lr_schedulers entry is not a ReduceLROnPlateau state (the YAML's lr_scheduler changed since it was written)")
        self.best, self.bad, self.cool = float(sd["best"]), int(sd["num_bad_epochs"]), int(sd["cooldown_counter"])
        self.last_epoch = int(sd.get("last_epoch", 0))


def _check_train_geometry(module: "TrainModule", data=None) -> None:
    """`fit` on a SpatialNet shape without training kernels fails here, before the first step, with the reason"""
    hp = getattr(module.arch, "hp", {})
    F = hp.get("num_freqs")
    if module.precision not in ("bf16-mixed", "bf16") and F is not None and F > 160:
        raise NotImplementedError(f"fit: the fp32-stream BACKWARD kernels hold whole frequency axes in LDS and stop at 160 bins (got num_freqs={F}); "
                                  "train 16-kHz models with trainer.precision=bf16-mixed")
    seg = getattr(data, "audio_time_len", None)
    if seg and getattr(data, "sr", None):
        frames = int(seg[0] * data.sr) // module.stft.n_hop + 1
        if frames > 256:
            raise NotImplementedError(f"fit: training segments of {seg[0]} s are {frames} frames; the training kernels keep one whole sequence per workgroup "
                                      "in LDS (<= 256 frames: 4 s at n_hop 128 / 8 kHz) — cut data.audio_time_len[0]")
    geo = (hp.get("dim_hidden"), hp.get("dim_ffn"), hp.get("dim_squeeze"), hp.get("num_heads"))
    if geo not in ((96, 192, 8, 4), (192, 384, 16, 4)):
        raise NotImplementedError("fit: the MI355X kernels are built for SpatialNet-small (dim_hidden 96, dim_ffn 192, dim_squeeze 8, 4 heads; configs/"
                                  "SpatialNet.yaml as shipped: fused training kernels) and SpatialNet-large (192 / 384 / 16 / 4: its \"for large\" comments; "
                                  f"generic backward, csrc/gbwd.hip); got {'/'.join(str(v) for v in geo)}")


def _is_fused_arch(cfg: dict) -> bool:
    """models.arch.SpatialNet.SpatialNet is served by the fused HIP step / forward-only path; every other arch (NBSS narrow-band
    models, OnlineSpatialNet) runs its torch.nn modules on the selected device through the generic loop"""
    arch = (cfg.get("model") or {}).get("arch")
    return isinstance(arch, dict) and arch.get("class_path") == "models.arch.SpatialNet.SpatialNet"


def _on_host(cfg: dict) -> bool:
    return cfg.get("trainer", {}).get("accelerator", "gpu") == "cpu" or not torch.cuda.is_available()


def fit(cfg: dict) -> Dict[str, Any]:
    tr = cfg.get("trainer", {})
    if _on_host(cfg) or not _is_fused_arch(cfg):
        # the narrow-band archs / OnlineSpatialNet on either device (BASELINE configs 1, 4, 5), and trainer.accelerator=cpu
        from nbss_amd.host_trainer import fit_generic
        return fit_generic(cfg, build_module, _instantiate)
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl")
    torch.manual_seed(int(cfg.get("seed_everything", 2)))
    module = build_module(cfg).to(dev)
    module.precision = str(tr.get("precision", "32"))
    data = _instantiate(cfg["data"]) if "data" in cfg else None
    if data is None:
        from data_loaders.synthetic import SyntheticDataModule
        data = SyntheticDataModule()
    if cfg.get("ckpt_path"):
        load_checkpoint(cfg["ckpt_path"], module)  # weights first: the engine binds the parameters below
    _check_train_geometry(module, data)
    eng, ts, gamma = _fused_step_for(module, cfg, dev)
    first_epoch = 0
    if cfg.get("ckpt_path"):
        first_epoch = load_checkpoint(cfg["ckpt_path"], module, ts, gamma if isinstance(gamma, _Plateau) else None) + 1  # (weights again: a no-op; now also the Adam and scheduler state)
        eng.version += 1
    ts.sync_replicas()  # rank 0's parameters / Adam state / step / lr on every rank (Lightning DDP's broadcast at fit start); no-op on one GPU
    ckpt_dir = tr.get("default_root_dir")
    log = []
    for epoch in range(first_epoch, int(tr.get("max_epochs", 1))):
        t0, n, tot = time.time(), 0, 0.0
        for x, ys, _ in data.batches(0, rank, world, epoch):
            loss = ts.step(x[:, module.channels].to(dev).contiguous(), ys[:, :, module.ref_channel].to(dev).contiguous())
            tot += float(loss)
            n += 1
        # validation pass of the epoch (forward-only path, every rank the same unsharded split): `val/neg_si_sdr` is what `val_metric: loss`
        # monitors in the reference (SharedTrainer.py:151-205) and what a ReduceLROnPlateau scheduler steps on
        vtot, vn = 0.0, 0
        for x, ys, _ in data.batches(1, 0, 1, 0):
            vl, _, _, _, _ = ts.forward_loss(x[:, module.channels].to(dev).contiguous(), ys[:, :, module.ref_channel].to(dev).contiguous(), need_grad=False)
            vtot, vn = vtot + float(vl), vn + 1
        val = vtot / vn if vn else float("nan")
        if isinstance(gamma, _Plateau):
            if vn:
                ts.lr = gamma.step(val, ts.lr)
        else:
            ts.lr *= gamma
        ts.check_replicas()  # every N steps (here: once per epoch): all ranks must still hold bitwise the same parameters and moments
        rec = {"epoch": epoch, "train/neg_si_sdr": tot / max(n, 1), "val/neg_si_sdr": val, "lr": ts.lr, "steps": n, "sec": time.time() - t0}
        log.append(rec)
        if rank == 0:
            print(json.dumps(rec), flush=True)
            if ckpt_dir:
                save_checkpoint(os.path.join(ckpt_dir, "checkpoints", "last.ckpt"), module, ts, epoch, global_step=ts.step_count,
                                plateau=gamma if isinstance(gamma, _Plateau) else None)
    if world > 1:
        torch.distributed.destroy_process_group()
    return {"log": log, "module": module}


def _setup(cfg: dict):
    """device, module, data module and a TrainStep for the non-training subcommands (inference path: no activations kept)"""
    tr = cfg.get("trainer", {})
    if tr.get("accelerator", "gpu") == "cpu" or not torch.cuda.is_available():
        raise RuntimeError("SharedTrainer: the SpatialNet path runs on MI355X HIP kernels only (no CPU path)")
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(int(cfg.get("seed_everything", 2)))
    module = build_module(cfg).to(dev)
    module.precision = str(tr.get("precision", "32"))
    if cfg.get("ckpt_path"):  # reference-format checkpoints: {"state_dict": {"arch.*": ...}}
        load_checkpoint(cfg["ckpt_path"], module)
    data = _instantiate(cfg["data"]) if "data" in cfg else None
    if data is None:
        from data_loaders.synthetic import SyntheticDataModule
        data = SyntheticDataModule()
    _, ts, _ = _fused_step_for(module, cfg, dev)
    return dev, module, data, ts


def _setup_generic(cfg: dict):
    """device, module, data module for the archs that run as torch.nn modules (PyTorch-ROCm compute on a HIP device, host otherwise)"""
    tr = cfg.get("trainer", {})
    dev = torch.device("cpu") if _on_host(cfg) else torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    torch.manual_seed(int(cfg.get("seed_everything", 2)))
    module = build_module(cfg).to(dev).eval()
    module.precision = str(tr.get("precision", "32"))
    if cfg.get("ckpt_path"):
        load_checkpoint(cfg["ckpt_path"], module)
    data = _instantiate(cfg["data"]) if "data" in cfg else None
    if data is None:
        from data_loaders.synthetic import SyntheticDataModule
        data = SyntheticDataModule()
    return dev, module, data


def _evaluate_generic(cfg: dict, stage: int) -> Dict[str, Any]:
    dev, module, data = _setup_generic(cfg)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    tot, tot_in, n = 0.0, 0.0, 0
    with torch.no_grad():
        for x, ys, _ in data.batches(stage, rank, world, 0):
            x, yr = x.to(dev), ys[:, :, module.ref_channel].to(dev).contiguous()
            yr_hat, _ = module.forward(x)
            loss, _, _ = module.loss(yr_hat=yr_hat, yr=yr, reorder=False, reduce_batch=True)
            mix = x[:, module.ref_channel][:, None].expand_as(yr).contiguous()
            loss_in, _, _ = module.loss(yr_hat=mix, yr=yr, reorder=False, reduce_batch=True)
            tot, tot_in, n = tot + float(loss), tot_in + float(loss_in), n + 1
    name = "val" if stage == 1 else "test"
    rec = {f"{name}/neg_si_sdr": tot / max(n, 1), f"{name}/si_sdr_improvement_dB": (tot_in - tot) / max(n, 1), "batches": n, "device": str(dev)}
    if rank == 0:
        print(json.dumps(rec), flush=True)
    return rec


def _predict_generic(cfg: dict) -> Dict[str, Any]:
    """`predict` for the torch.nn archs.  OnlineSpatialNet with a fixed-size state ('mhsa(N)', 'ret(..)' without rotary positions) is
    evaluated the way it is deployed: causal, `stream_chunk` frames at a time (CLI: --stream_chunk N, default 8) through
    OnlineStreamer, whose step is captured once into a HIP graph on a HIP device and replayed (BASELINE config 5)."""
    from models.arch.OnlineSpatialNet import OnlineSpatialNet
    dev, module, data = _setup_generic(cfg)
    out_dir = cfg.get("trainer", {}).get("default_root_dir")
    chunk = int(cfg.get("stream_chunk", 8))
    outs, info = [], {"device": str(dev), "streamed": False, "graph_replays": 0}
    with torch.no_grad():
        for bi, (x, ys, paras) in enumerate(data.batches(2, 0, 1, 0)):
            x = x.to(dev)
            if isinstance(module.arch, OnlineSpatialNet) and chunk > 0:
                yr_hat, st = module.forward_streaming(x, chunk)
                info["streamed"], info["graph_replays"] = True, info["graph_replays"] + st["graph_replays"]
                info["frames_per_s"], info["native"] = st["frames_per_s"], st["native"]
            else:
                yr_hat, _ = module.forward(x)
            outs.append(yr_hat.cpu())
            if out_dir:
                os.makedirs(out_dir, exist_ok=True)
                torch.save({"yr_hat": outs[-1], "paras": paras}, os.path.join(out_dir, f"predict_{bi:05d}.pt"))
    return {"yr_hat": outs, **info}


def evaluate(cfg: dict, stage: int) -> Dict[str, Any]:
    """`validate` (stage 1) / `test` (stage 2): uPIT neg-SI-SDR of the separated signals and the SI-SDR improvement over the
    unprocessed reference-channel mixture, through the forward-only path (SharedTrainer.py:151-205 without the PESQ/STOI pools)."""
    if not _is_fused_arch(cfg):
        return _evaluate_generic(cfg, stage)
    from nbss_amd import ops
    dev, module, data, ts = _setup(cfg)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    tot, tot_in, n = 0.0, 0.0, 0
    for x, ys, _ in data.batches(stage, rank, world, 0):
        xs = x[:, module.channels].to(dev).contiguous()
        yr = ys[:, :, module.ref_channel].to(dev).contiguous()
        loss, yr_hat, _, _, _ = ts.forward_loss(xs, yr, need_grad=False)
        mix = xs[:, module.channels.index(module.ref_channel)][:, None].expand_as(yr).contiguous()
        loss_in, _, _ = ops.pit_neg_sisdr(ts.lib, mix, yr, need_grad=False)
        tot += float(loss)
        tot_in += float(loss_in)
        n += 1
    name = "val" if stage == 1 else "test"
    rec = {f"{name}/neg_si_sdr": tot / max(n, 1), f"{name}/si_sdr_improvement_dB": (tot_in - tot) / max(n, 1), "batches": n}
    if rank == 0:
        print(json.dumps(rec), flush=True)
    return rec


def predict(cfg: dict) -> Dict[str, Any]:
    """`predict`: separated waveforms [B,Spk,N] of the test split (returned; written as .pt files when trainer.default_root_dir is set)"""
    if not _is_fused_arch(cfg):
        return _predict_generic(cfg)
    dev, module, data, ts = _setup(cfg)
    out_dir = cfg.get("trainer", {}).get("default_root_dir")
    outs = []
    for bi, (x, ys, paras) in enumerate(data.batches(2, 0, 1, 0)):
        xs = x[:, module.channels].to(dev).contiguous()
        yr = ys[:, :, module.ref_channel].to(dev).contiguous()
        _, yr_hat, _, _, perm = ts.forward_loss(xs, yr, need_grad=False)
        outs.append(yr_hat.cpu())
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
            torch.save({"yr_hat": outs[-1], "paras": paras}, os.path.join(out_dir, f"predict_{bi:05d}.pt"))
    return {"yr_hat": outs}


class TrainCLI:
    """`TrainCLI(TrainModule, ...)`-shaped entry point (reference :344-371)"""

    def __init__(self, *args, argv: Optional[List[str]] = None, **kwargs):
        sub, cfg = parse_cli(sys.argv[1:] if argv is None else argv)
        self.subcommand, self.config = sub, cfg
        if sub == "fit":
            self.result = fit(cfg)
        elif sub in ("validate", "test"):
            self.result = evaluate(cfg, 1 if sub == "validate" else 2)
        else:
            self.result = predict(cfg)


if __name__ == "__main__":
    TrainCLI(TrainModule)
