"""STFT / iSTFT — drop-in for the reference's models/io/stft.py (STFT(n_fft, n_hop, win_len=None, win='hann_window'),
.stft(x) -> (complex [..., F, T], length), .istft(X, original_len)).  Tensors on a HIP device are transformed by the MI355X
DFT-as-GEMM kernels (nbss_amd/csrc/signal.hip, fp32; n_fft in {256, 512}, n_hop = n_fft/2, win_len = n_fft — every shipped
config); TrainModule's fused path uses the stft+norm / inorm+istft entry points directly.  Host tensors (`trainer.accelerator=cpu`,
the plumbing run of the narrow-band models) go through torch.stft / torch.istft with the reference's arguments (stft.py:49-97)."""
from typing import Any, Optional, Tuple

import torch
from torch import Tensor, nn

paras_16k = {"n_fft": 512, "n_hop": 256, "win_len": 512}
paras_8k = {"n_fft": 256, "n_hop": 128, "win_len": 256}


class _IstftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, Xr, length):
        from nbss_amd import ops
        from nbss_amd._lib import hip
        ones = torch.ones(Xr.shape[:3], dtype=torch.float32, device=Xr.device)
        ctx.mod, ctx.ones = mod, ones
        return ops.inorm_istft_fwd(hip(), mod.n_fft, mod._tables(Xr.device), Xr, ones, length)

    @staticmethod
    def backward(ctx, dy):
        from nbss_amd import ops
        from nbss_amd._lib import hip
        return None, ops.inorm_istft_bwd(hip(), ctx.mod.n_fft, ctx.mod._tables(dy.device), dy.contiguous(), ctx.ones), None


class STFT(nn.Module):
    def __init__(self, n_fft: int, n_hop: int, win_len: Optional[int] = None, win: str = "hann_window") -> None:
        super().__init__()
        self.n_fft, self.n_hop, self.win_len = n_fft, n_hop, win_len if win_len is not None else n_fft
        assert win in ("hann_window", "sqrt_hann_window"), win
        self.hip_ok = self.n_hop * 2 == self.n_fft and self.win_len == self.n_fft and n_fft in (256, 512)
        self.win = win
        self.repr = str((n_fft, n_hop, win, win_len))
        w = torch.hann_window(self.n_fft)  # (n_fft, not win_len: the reference's buffer shape, stft.py:29-35; `stft.window` is a checkpoint key)
        self.register_buffer("window", w if win == "hann_window" else w.sqrt())
        self._tab = {}

    def _tables(self, device):
        if not self.hip_ok:
            raise NotImplementedError("MI355X STFT kernels cover n_fft in {256,512}, n_hop = n_fft/2, win_len = n_fft")
        from nbss_amd import ops
        from nbss_amd._lib import hip
        key = str(device)
        if key not in self._tab:
            self._tab[key] = ops.stft_tables(hip(), self.n_fft, 0 if self.win == "hann_window" else 1, device)
        return self._tab[key]

    def forward(self, X: Tensor, original_len: int = None, inverse=False) -> Any:
        return self.istft(X, original_len) if inverse else self.stft(X)

    def stft(self, x: Tensor) -> Tuple[Tensor, int]:
        from nbss_amd import ops
        from nbss_amd._lib import NBSS_F32, hip
        shape = list(x.shape)
        if not x.is_cuda or not self.hip_ok:  # host tensors, and STFT geometries outside the HIP kernels' {256, 512} / hop = n_fft/2
            X = torch.stft(x.reshape(-1, shape[-1]).float(), n_fft=self.n_fft, hop_length=self.n_hop, win_length=self.win_len,
                           window=self.window.to(x.device), return_complex=True)
            return X.reshape(shape[:-1] + list(X.shape[-2:])), shape[-1]
        x2 = x.reshape(-1, 1, shape[-1]).float().contiguous()
        Xn, mm = ops.stft_norm_fwd(hip(), self.n_fft, NBSS_F32, self._tables(x.device), x2, 0)  # [B',F,T,2], |X|+1e-6
        X = torch.view_as_complex((Xn * mm[..., None]).contiguous())  # undo the fused per-bin normalisation
        return X.reshape(shape[:-1] + list(X.shape[-2:])), shape[-1]

    def istft(self, X: Tensor, original_len: int = None) -> Tensor:
        shape = list(X.shape)
        if not X.is_cuda or not self.hip_ok:
            Xf = X.reshape(-1, *shape[-2:]).to(torch.complex64)
            y = torch.istft(Xf, n_fft=self.n_fft, hop_length=self.n_hop, win_length=self.win_len, window=self.window.to(X.device), length=original_len)
            return y.reshape(shape[:-2] + [original_len])
        Xr = torch.view_as_real(X.reshape(-1, *shape[-2:]).to(torch.complex64)).contiguous()  # [B',F,T,2]
        y = _IstftFn.apply(self, Xr, int(original_len))
        return y.reshape(shape[:-2] + [original_len])

    def extra_repr(self) -> str:
        return self.repr

    def _load_from_state_dict(self, *args, **kwargs):  # the window is a constant, like in the reference (stft.py:102)
        return
