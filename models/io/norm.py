"""Norm — drop-in for the reference's models/io/norm.py (Norm(mode, online).norm / .inorm on complex [B,C,F,T]).
The hot configuration ('frequency', online=True: per-T-F-bin magnitude of the reference channel) is fused into the
STFT / iSTFT kernels by SharedTrainer.TrainModule; this class offers the same API for stand-alone use with cheap
elementwise torch ops on the device tensors ('forgetting' is not provided)."""
from typing import Any, Literal, Optional, Tuple

import torch
from torch import Tensor, nn


class Norm(nn.Module):
    def __init__(self, mode: Optional[Literal["utterance", "frequency", "forgetting", "none"]], online: bool = True) -> None:
        super().__init__()
        if mode == "forgetting":
            raise NotImplementedError("Norm('forgetting') is not on the SpatialNet path")
        self.mode, self.online = mode, online

    def forward(self, X: Tensor, norm_paras: Any = None, inverse: bool = False) -> Any:
        return self.inorm(X, norm_paras) if inverse else self.norm(X, norm_paras=norm_paras)

    def norm(self, X: Tensor, norm_paras: Any = None, ref_channel: int = None, eps: float = 1e-6) -> Tuple[Tensor, Any]:
        if self.mode in ("none", None):
            return X, (X[:, [ref_channel]].clone(), None)
        if norm_paras is None:
            Xr = X[:, [ref_channel]].clone()
            mag = Xr.abs()
            if self.mode == "frequency":
                XrMM = mag + eps if self.online else mag.mean(dim=3, keepdim=True) + eps
            else:
                assert self.mode == "utterance", self.mode
                XrMM = mag.mean(dim=(2,) if self.online else (2, 3), keepdim=True) + eps
        else:
            Xr, XrMM = norm_paras
        X[:, :, :, :] /= XrMM
        return X, (Xr, XrMM)

    def inorm(self, X: Tensor, norm_paras: Any) -> Tensor:
        Xr, XrMM = norm_paras
        return X * XrMM

    def extra_repr(self) -> str:
        return f"{self.mode}, online={self.online}"
