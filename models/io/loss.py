"""Loss — drop-in for the reference's models/io/loss.py: `neg_si_sdr` (the callable the YAML names,
configs/SpatialNet.yaml:38) and `Loss(loss_func, pit, loss_func_kwargs).forward(yr_hat, yr, reorder, reduce_batch)
-> (loss, perms, yr_hat)`, `to_CC`.  neg-SI-SDR with or without PIT runs on the MI355X kernel
(nbss_amd/csrc/loss_optim.hip, which restates torchmetrics' si_sdr / pit); the other loss functions of the
reference (neg_sa_sdr, neg_snr, cirm_mse, cc_mse) are not on the SpatialNet path and raise."""
from typing import Any, Callable, Dict, Tuple

import torch
from torch import Tensor, nn


class _PitSiSdrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, target, pit: bool):
        from nbss_amd import ops
        from nbss_amd._lib import hip
        B, S, N = preds.shape
        p, t = preds.float().contiguous(), target.float().contiguous()
        if pit:
            loss, perm, dp = ops.pit_neg_sisdr(hip(), p, t, need_grad=True)
            # per-item losses for reduce_batch=False: recompute cheaply from the permutation
            losses = None
        else:  # identity pairing == PIT over one speaker at a time
            parts = [ops.pit_neg_sisdr(hip(), p[:, s:s + 1].contiguous(), t[:, s:s + 1].contiguous(), need_grad=True) for s in range(S)]
            loss = sum(x[0] for x in parts) / S
            dp = torch.cat([x[2] for x in parts], 1) / S
            perm = torch.arange(S, device=p.device, dtype=torch.int32).expand(B, S).contiguous()
        ctx.save_for_backward(dp)
        ctx.mark_non_differentiable(perm)
        return loss.reshape(()), perm

    @staticmethod
    def backward(ctx, dloss, _):
        (dp,) = ctx.saved_tensors
        return dp * dloss, None, None


def neg_si_sdr(preds: Tensor, target: Tensor) -> Tensor:
    """-mean over speakers of SI-SDR, shape [batch] (loss.py:21-29)."""
    B, S = preds.shape[:2]
    from nbss_amd import ops
    from nbss_amd._lib import hip
    out = []
    for b in range(B):  # per-item values through the same kernel (identity pairing)
        vals = [ops.pit_neg_sisdr(hip(), preds[b:b + 1, s:s + 1].float().contiguous(), target[b:b + 1, s:s + 1].float().contiguous(), need_grad=False)[0]
                for s in range(S)]
        out.append(sum(vals) / S)
    return torch.cat(out)


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
        if not reduce_batch:
            raise NotImplementedError("reduce_batch=False is only used by the reference's test step")
        loss, perm = _PitSiSdrFn.apply(yr_hat, yr, self.pit)
        perms = perm.long() if self.pit else None
        if reorder and perms is not None:
            yr_hat = torch.gather(yr_hat, 1, perms[..., None].expand_as(yr_hat))  # torchmetrics pit_permutate
        return loss, perms, yr_hat

    def to_CC(self, out: Tensor, Xr: Tensor, stft, XrMM: Tensor):
        return out, {"out": out, "Xr": Xr, "stft": stft, "XrMM": XrMM}

    def extra_repr(self) -> str:
        return f"loss_func={self.loss_func.__name__}(), pit={self.pit}, mask={self.mask}"
