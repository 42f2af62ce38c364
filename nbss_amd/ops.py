"""Tensor-level wrappers of the C ABI (include/nbss_hip.h): torch is used for device memory
and streams only.  Every function takes the library handle explicitly (`lib`); the product
passes `nbss_amd._lib.hip()`; tests may pass the host-emulator build to run the same kernel
sources on CPU tensors.  There is no other code path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch
from torch import Tensor

from ._lib import NBSS_BF16, NBSS_F32, Cfg, Lib, NbssError
from .params import param_table


def _is_emu(lib: Lib) -> bool:
    if not hasattr(lib, "_is_emu"):
        lib._is_emu = "emulator" in lib.build_info()
    return lib._is_emu


def stream_dtype(cfg: Cfg) -> torch.dtype:
    return torch.bfloat16 if cfg.dtype == NBSS_BF16 else torch.float32


def _ptr(lib: Lib, t: Optional[Tensor], dtype=None) -> Optional[int]:
    if t is None:
        return None
    if not t.is_contiguous():
        raise NbssError("nbss_amd ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise NbssError(f"expected {dtype}, got {t.dtype}")
    if _is_emu(lib):
        if t.device.type != "cpu":
            raise NbssError("the emulator library only takes CPU tensors")
    elif t.device.type != "cuda":
        raise NbssError("libnbss_hip.so needs tensors on a HIP device (no CPU fallback exists)")
    return t.data_ptr()


def _stream(lib: Lib, t: Tensor) -> Optional[int]:
    if _is_emu(lib):
        return None
    return torch.cuda.current_stream(t.device).cuda_stream


def random_params(lib: Lib, cfg: Cfg, device, seed: int = 0) -> Tensor:
    """flat fp32 parameter buffer with plausible random values (tools / micro-benchmarks: weights ~ N(0, fan_in^-1/2), norm
    weights 1, biases small) — no state_dict needed"""
    n = lib.nbss_param_count(C.byref(cfg))
    if n <= 0:
        raise NbssError("unsupported configuration")
    g = torch.Generator().manual_seed(seed)
    flat = torch.zeros(n, dtype=torch.float32)
    for name, (off, shape) in param_table(lib, cfg).items():
        k = 1
        for d in shape:
            k *= d
        if len(shape) == 1:
            is_norm_w = name.endswith("weight") and ("norm" in name or name.split(".")[-2] in ("0", "6"))
            v = torch.ones(k) if is_norm_w else (torch.full((k,), 0.25) if len(shape) == 1 and name.endswith(".2.weight") else 0.02 * torch.randn(k, generator=g))
        else:
            fan_in = k // shape[0]
            v = torch.randn(k, generator=g) / fan_in ** 0.5
        flat[off:off + k] = v
    return flat.to(device)


def flatten_params(lib: Lib, cfg: Cfg, p: Dict[str, Tensor], device) -> Tensor:
    """state_dict-style dict -> flat fp32 buffer in nbss_param_table order."""
    n = lib.nbss_param_count(C.byref(cfg))
    if n <= 0:
        raise NbssError("unsupported configuration")
    flat = torch.zeros(n, dtype=torch.float32)
    for name, (off, shape) in param_table(lib, cfg).items():
        t = p[name]
        if tuple(t.shape) != tuple(shape):
            raise NbssError(f"{name}: shape {tuple(t.shape)} != {shape}")
        flat[off:off + t.numel()] = t.detach().reshape(-1).to(torch.float32).cpu()
    return flat.to(device)


def pack_params(lib: Lib, cfg: Cfg, flat: Tensor, out: Optional[Tensor] = None) -> Tensor:
    nbytes = lib.nbss_packed_bytes(C.byref(cfg))
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=flat.device)
    lib.call("nbss_pack_params", C.byref(cfg), _ptr(lib, flat, torch.float32), _ptr(lib, out), _stream(lib, flat))
    return out


def encoder_fwd(lib, cfg, flat, packed, xin):
    y = torch.empty(cfg.B, cfg.F, cfg.T, cfg.H, dtype=stream_dtype(cfg), device=xin.device)
    lib.call("nbss_encoder_fwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, packed), _ptr(lib, xin, stream_dtype(cfg)), _ptr(lib, y), _stream(lib, xin))
    return y


def decoder_fwd(lib, cfg, flat, packed, x):
    out = torch.empty(cfg.B, cfg.F, cfg.T, cfg.C_out, dtype=torch.float32, device=x.device)
    lib.call("nbss_decoder_fwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, packed), _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, out), _stream(lib, x))
    return out


def fconv_fwd(lib, cfg, flat, packed, layer, which, x):
    y = torch.empty_like(x)
    lib.call("nbss_fconv_fwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, packed), layer, which, _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, y), _stream(lib, x))
    return y


def full_fwd(lib, cfg, flat, packed, layer, x):
    y = torch.empty_like(x)
    lib.call("nbss_full_fwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, packed), layer, _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, y), _stream(lib, x))
    return y


def scratch(nbytes: int, device) -> Tensor:
    """caller-owned scratch for the library (workspaces, saved activations).  NBSS_POISON_SCRATCH=1 (set by tests/conftest.py)
    fills it with 0xFF bytes — NaN in bf16 and fp32 — so that a kernel that reads scratch it has not written shows up as NaN in
    the checked outputs instead of passing on freshly mapped, zeroed pages."""
    t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    if os.environ.get("NBSS_POISON_SCRATCH") == "1":
        t.fill_(0xFF)
    return t


def mhsa_save(lib, cfg, device) -> Tensor:
    """buffer for what mhsa_bwd needs from the forward pass (attention output before out_proj + log-sum-exp rows)"""
    return scratch(lib.nbss_mhsa_save_bytes(C.byref(cfg)), device)


def mhsa_fwd(lib, cfg, flat, packed, layer, x, o_save=None):
    y = torch.empty_like(x)
    lib.call("nbss_mhsa_fwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, packed), layer, _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, y),
             _ptr(lib, o_save, torch.uint8 if o_save is not None else None), _stream(lib, x))
    return y


def mhsa_bwd(lib, cfg, flat, grads, packed, layer, x, dy, o_save, ws):
    dx = torch.empty_like(x)
    lib.call("nbss_mhsa_bwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, grads, torch.float32), _ptr(lib, packed), layer,
             _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, dy, stream_dtype(cfg)), _ptr(lib, o_save, torch.uint8), _ptr(lib, dx), _ptr(lib, ws),
             _stream(lib, x))
    return dx


def tconvffn_save(lib, cfg, device) -> Optional[Tensor]:
    """buffer for what a training-mode T-ConvFFN forward keeps for backward (None: this geometry / stream type recomputes instead)"""
    n = lib.nbss_tconvffn_save_bytes(C.byref(cfg))
    return scratch(n, device) if n > 0 else None


def tconvffn_fwd(lib, cfg, flat, packed, layer, x, t_save=None):
    y = torch.empty_like(x)
    lib.call("nbss_tconvffn_fwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, packed), layer, _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, y),
             _ptr(lib, t_save, torch.uint8) if t_save is not None else None, _stream(lib, x))
    return y


def workspace(lib, cfg, device) -> Tensor:
    return scratch(lib.nbss_workspace_bytes(C.byref(cfg)), device)


def tconvffn_bwd(lib, cfg, flat, grads, packed, layer, x, dy, ws, t_save=None):
    dx = torch.empty_like(x)
    lib.call("nbss_tconvffn_bwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, grads, torch.float32), _ptr(lib, packed), layer,
             _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, dy, stream_dtype(cfg)), _ptr(lib, t_save, torch.uint8) if t_save is not None else None,
             _ptr(lib, dx), _ptr(lib, ws), _stream(lib, x))
    return dx


def fconv_bwd(lib, cfg, flat, grads, packed, layer, which, x, dy, ws):
    dx = torch.empty_like(x)
    lib.call("nbss_fconv_bwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, grads, torch.float32), _ptr(lib, packed), layer, which,
             _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, dy, stream_dtype(cfg)), _ptr(lib, dx), _ptr(lib, ws), _stream(lib, x))
    return dx


def full_bwd(lib, cfg, flat, grads, packed, layer, x, dy, ws):
    dx = torch.empty_like(x)
    lib.call("nbss_full_bwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, grads, torch.float32), _ptr(lib, packed), layer,
             _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, dy, stream_dtype(cfg)), _ptr(lib, dx), _ptr(lib, ws), _stream(lib, x))
    return dx


def decoder_bwd(lib, cfg, flat, grads, packed, x, dout, ws):
    dx = torch.empty_like(x)
    lib.call("nbss_decoder_bwd", C.byref(cfg), _ptr(lib, flat), _ptr(lib, grads, torch.float32), _ptr(lib, packed),
             _ptr(lib, x, stream_dtype(cfg)), _ptr(lib, dout, torch.float32), _ptr(lib, dx), _ptr(lib, ws), _stream(lib, x))
    return dx


def encoder_bwd(lib, cfg, grads, xin, dy):
    lib.call("nbss_encoder_bwd", C.byref(cfg), _ptr(lib, grads, torch.float32), _ptr(lib, xin, stream_dtype(cfg)),
             _ptr(lib, dy, stream_dtype(cfg)), _stream(lib, xin))


def stft_tables(lib, n_fft: int, window: int, device) -> Tensor:
    nb = lib.nbss_stft_tables_bytes(n_fft)
    if nb <= 0:
        raise NbssError(f"n_fft={n_fft} is not supported (256 or 512)")
    tab = torch.empty(nb // 4, dtype=torch.float32, device=device)
    lib.call("nbss_stft_tables", n_fft, window, _ptr(lib, tab), _stream(lib, tab))
    return tab


def stft_norm_fwd(lib, n_fft, dtype, tables, x, ref_channel):
    """x [B,C,N] fp32 -> (X [B,F,T,2C] stream dtype, xrmm [B,F,T] fp32)"""
    B, Cc, N = x.shape
    F, T = n_fft // 2 + 1, N // (n_fft // 2) + 1
    X = torch.empty(B, F, T, 2 * Cc, dtype=torch.bfloat16 if dtype == NBSS_BF16 else torch.float32, device=x.device)
    xrmm = torch.empty(B, F, T, dtype=torch.float32, device=x.device)
    lib.call("nbss_stft_norm_fwd", n_fft, dtype, B, Cc, N, ref_channel, _ptr(lib, tables), _ptr(lib, x, torch.float32), _ptr(lib, X), _ptr(lib, xrmm),
             _stream(lib, x))
    return X, xrmm


def inorm_istft_fwd(lib, n_fft, tables, out, xrmm, N):
    """out [B,F,T,2S] fp32 -> y [B,S,N] fp32"""
    B, F, T, S2 = out.shape
    S = S2 // 2
    ws = torch.empty(lib.nbss_istft_ws_bytes(n_fft, B, S, N) // 4, dtype=torch.float32, device=out.device)
    y = torch.empty(B, S, N, dtype=torch.float32, device=out.device)
    lib.call("nbss_inorm_istft_fwd", n_fft, B, S, N, _ptr(lib, tables), _ptr(lib, out, torch.float32), _ptr(lib, xrmm, torch.float32), _ptr(lib, ws),
             _ptr(lib, y), _stream(lib, out))
    return y


def inorm_istft_bwd(lib, n_fft, tables, dy, xrmm):
    B, S, N = dy.shape
    F, T = xrmm.shape[1], xrmm.shape[2]
    dout = torch.empty(B, F, T, 2 * S, dtype=torch.float32, device=dy.device)
    lib.call("nbss_inorm_istft_bwd", n_fft, B, S, N, _ptr(lib, tables), _ptr(lib, dy, torch.float32), _ptr(lib, xrmm, torch.float32), _ptr(lib, dout),
             _stream(lib, dy))
    return dout


def pit_neg_sisdr(lib, preds, target, need_grad=True, return_items=False):
    """-> (loss [1], perm [B,S] int32, dpreds or None[, per-item losses [B]])"""
    B, S, N = preds.shape
    dev = preds.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    perm = torch.empty(B, S, dtype=torch.int32, device=dev)
    dp = torch.empty_like(preds) if need_grad else None
    ws = torch.empty(lib.nbss_pit_ws_bytes(B, S) // 4, dtype=torch.float32, device=dev)
    lib.call("nbss_pit_neg_sisdr", B, S, N, _ptr(lib, preds, torch.float32), _ptr(lib, target, torch.float32), _ptr(lib, loss), _ptr(lib, perm),
             _ptr(lib, dp), _ptr(lib, ws), _stream(lib, preds))
    if return_items:  # workspace tail (include/nbss_hip.h): ... | per-item loss [B] | pairing coefficients [3 B S]
        return loss, perm, dp, ws[ws.numel() - B - 3 * B * S: ws.numel() - 3 * B * S].clone()
    return loss, perm, dp


def clip_adam_step(lib, params, grads, exp_avg, exp_avg_sq, scratch, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                   max_norm=5.0, grad_scale=1.0, zero_grad=True, decoupled_weight_decay=False):
    lib.call("nbss_clip_adam_step", params.numel(), _ptr(lib, params, torch.float32), _ptr(lib, grads, torch.float32), _ptr(lib, exp_avg, torch.float32),
             _ptr(lib, exp_avg_sq, torch.float32), _ptr(lib, scratch, torch.float32), float(max_norm), float(grad_scale), float(lr), float(betas[0]),
             float(betas[1]), float(eps), float(weight_decay), int(step), int(bool(zero_grad)) | (2 if decoupled_weight_decay else 0), _stream(lib, params))


def clip_adam_step_dev(lib, params, grads, exp_avg, exp_avg_sq, scratch, hyper, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=5.0,
                       grad_scale=1.0, zero_grad=True, decoupled_weight_decay=False):
    """clip_adam_step with the per-step scalars read from the DEVICE buffer `hyper` = [lr, 1 - beta1^step, sqrt(1 - beta2^step)] (filled by
    nbss_adam_hyper on the host): the launch sequence has no per-step arguments and can be replayed from a HIP graph"""
    lib.call("nbss_clip_adam_step_dev", params.numel(), _ptr(lib, params, torch.float32), _ptr(lib, grads, torch.float32), _ptr(lib, exp_avg, torch.float32),
             _ptr(lib, exp_avg_sq, torch.float32), _ptr(lib, scratch, torch.float32), _ptr(lib, hyper, torch.float32), float(max_norm), float(grad_scale),
             float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(bool(zero_grad)) | (2 if decoupled_weight_decay else 0), _stream(lib, params))


def selftest_mma(lib, dtype: int, kperm: int, A: Tensor, B: Tensor) -> Tensor:
    D = torch.empty(16, 16, dtype=torch.float32, device=A.device)
    lib.call("nbss_selftest_mma", dtype, kperm, _ptr(lib, A, torch.float32), _ptr(lib, B, torch.float32), _ptr(lib, D), _stream(lib, A))
    return D


def graph_guard_save(ctx, runner, saved, params):
    """state of a native autograd.Function between forward and backward: the runner, its module (kept alive while the graph is), the saved device
    tensors and the version counters of the parameters backward will re-read from the live module"""
    ctx.runner, ctx.net, ctx.saved, ctx.params = runner, runner.net, saved, params
    ctx.versions = [p._version for p in params]


def graph_guard_check(ctx, what: str):
    """torch.nn raises when a tensor saved for backward was modified in place or the graph was already freed; the native paths re-read the module's
    parameters in backward, so they check the same two conditions themselves"""
    if ctx.saved is None:
        raise RuntimeError(f"{what}: trying to backward through the graph a second time: the native path frees its saved state after the first backward "
                           "(retain_graph is not supported)")
    for p, v in zip(ctx.params, ctx.versions):
        if p._version != v:
            raise RuntimeError(f"{what}: a parameter needed for the gradient computation was modified in place between forward and backward "
                               f"(shape {tuple(p.shape)}, version {p._version}, expected {v})")
