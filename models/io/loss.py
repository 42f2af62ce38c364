"""Loss — drop-in for the reference's models/io/loss.py: `neg_si_sdr` (the callable the YAML names,
configs/SpatialNet.yaml:38) and `Loss(loss_func, pit, loss_func_kwargs).forward(yr_hat, yr, reorder, reduce_batch)
-> (loss, perms, yr_hat)`, `to_CC`.  For tensors on a HIP device neg-SI-SDR with or without PIT runs on the MI355X kernel
(nbss_amd/csrc/loss_optim.hip, which restates torchmetrics' si_sdr / pit); host tensors (`trainer.accelerator=cpu`) go through
the same closed forms in torch.  The other loss functions of the reference (neg_sa_sdr, neg_snr, cirm_mse, cc_mse) are not on
the SpatialNet path and raise."""
import itertools
from typing import Any, Callable, Dict, Tuple

import torch
from torch import Tensor, nn


def _identity_pairing(p: Tensor, t: Tensor, need_grad: bool):
    """no PIT: speaker s of the estimate is paired with speaker s of the target = the PIT kernel on one speaker at a time
    (S launches, not B*S); returns (mean loss [1], per-item loss [B], d loss / d p or None)"""
    from nbss_amd import ops
    from nbss_amd._lib import hip
    S = p.shape[1]
    parts = [ops.pit_neg_sisdr(hip(), p[:, s:s + 1].contiguous(), t[:, s:s + 1].contiguous(), need_grad=need_grad, return_items=True) for s in range(S)]
    loss = sum(x[0] for x in parts) / S
    items = sum(x[3] for x in parts) / S
    dp = torch.cat([x[2] for x in parts], 1) / S if need_grad else None
    return loss, items, dp


class _PitSiSdrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, target, pit: bool):
        from nbss_amd import ops
        from nbss_amd._lib import hip
        B, S, N = preds.shape
        p, t = preds.float().contiguous(), target.float().contiguous()
        if pit:
            loss, perm, dp, items = ops.pit_neg_sisdr(hip(), p, t, need_grad=True, return_items=True)
        else:
            loss, items, dp = _identity_pairing(p, t, True)
            perm = torch.arange(S, device=p.device, dtype=torch.int32).expand(B, S).contiguous()
        ctx.save_for_backward(dp)
        ctx.mark_non_differentiable(perm, items)
        return loss.reshape(()), perm, items

    @staticmethod
    def backward(ctx, dloss, _p, _i):
        (dp,) = ctx.saved_tensors
        return dp * dloss, None, None


def _host_pair_sisdr(p: Tensor, t: Tensor) -> Tensor:
    """[B,S,N] x [B,S,N] -> SI-SDR of every (estimate i, target j) pair [B,S,S] (zero_mean=False; eps = float32 machine epsilon)"""
    eps = torch.finfo(p.dtype).eps
    pt = torch.einsum("bin,bjn->bij", p, t)
    tt, pp = (t * t).sum(-1)[:, None, :], (p * p).sum(-1)[:, :, None]
    alpha = (pt + eps) / (tt + eps)
    num = alpha * alpha * tt
    return 10 * torch.log10((num + eps) / (num - 2 * alpha * pt + pp + eps))


def _host_pit(p: Tensor, t: Tensor, pit: bool):
    """host (torch, differentiable) uPIT neg-SI-SDR: per-item loss [B] and the pairing perm [B,S] (perm[s] = estimate paired with target s)"""
    B, S, _ = p.shape
    sd = _host_pair_sisdr(p.float(), t.float())
    perms = list(itertools.permutations(range(S))) if pit else [tuple(range(S))]
    vals = torch.stack([-sum(sd[:, pm[s], s] for s in range(S)) / S for pm in perms], 1)  # [B, S!] in itertools order (torchmetrics' order)
    best, idx = vals.min(1)
    return best, torch.tensor(perms, dtype=torch.long, device=p.device)[idx]


def neg_si_sdr(preds: Tensor, target: Tensor) -> Tensor:
    """-mean over speakers of SI-SDR, shape [batch] (loss.py:21-29)"""
    if not preds.is_cuda:
        return _host_pit(preds, target, False)[0]
    return _identity_pairing(preds.float().contiguous(), target.float().contiguous(), False)[1]


def _unsupported(name):
    def f(*a, **k):
        raise NotImplementedError(f"{name} is not on the MI355X SpatialNet path (only neg_si_sdr is)")
    f.__name__ = name
    return f


neg_sa_sdr, neg_snr, cirm_mse, cc_mse = (_unsupported(n) for n in ("neg_sa_sdr", "neg_snr", "cirm_mse", "cc_mse"))


class Loss(nn.Module):
    is_scale_invariant_loss: bool
    name: str
    mask: str

    def __init__(self, loss_func: Callable, pit: bool, loss_func_kwargs: Dict[str, Any] = dict()):
        super().__init__()
        if isinstance(loss_func, str):  # YAML callable path
            import importlib
            mod, _, fn = loss_func.rpartition(".")
            loss_func = getattr(importlib.import_module(mod), fn)
        if loss_func is not neg_si_sdr:
            raise NotImplementedError(f"Loss({getattr(loss_func, '__name__', loss_func)}): only neg_si_sdr has an MI355X kernel")
        self.loss_func, self.pit, self.loss_func_kwargs = loss_func, pit, loss_func_kwargs
        self.is_scale_invariant_loss = True
        self.name = loss_func.__name__
        self.mask = None

    def forward(self, yr_hat: Tensor, yr: Tensor, reorder: bool = None, reduce_batch: bool = True, **kwargs) -> Tuple[Tensor, Tensor, Tensor]:
        if not yr_hat.is_cuda:  # host path
            items, perm = _host_pit(yr_hat, yr, self.pit)
            loss = items.mean()
        else:
            loss, perm, items = _PitSiSdrFn.apply(yr_hat, yr, self.pit)
        if not reduce_batch:  # the reference's test step: one loss per utterance (loss.py:111-118), no gradient
            loss = items
        perms = perm.long() if self.pit else None
        if reorder and perms is not None:
            yr_hat = torch.gather(yr_hat, 1, perms[..., None].expand_as(yr_hat))  # torchmetrics pit_permutate
        return loss, perms, yr_hat

    def to_CC(self, out: Tensor, Xr: Tensor, stft, XrMM: Tensor):
        return out, {"out": out, "Xr": Xr, "stft": stft, "XrMM": XrMM}

    def extra_repr(self) -> str:
        return f"loss_func={self.loss_func.__name__}(), pit={self.pit}, mask={self.mask}"
