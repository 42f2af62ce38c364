"""Native NBC inference (nbss_amd/nbc.py over nbss_nb_attention_relpos_fwd / nbss_nb_group_norm / the tap-GEMM building blocks) against the torch.nn module
it reads its parameters from (models/arch/NBC.py, pinned to the reference's NBC by tests/test_nb_models.py), and the two new building blocks against torch."""
import math
import os

import pytest
import torch
import torch.nn.functional as Fn

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, NBSS_F32
from util import rel_l2


def _net(hidden, heads, ffn, layers=2, din=4, dout=4):
    from models.arch.NBC import NBC
    torch.manual_seed(3)
    net = NBC(dim_input=din, dim_output=dout, n_layers=layers, encoder_kernel_size=4, n_heads=heads, hidden_size=hidden, ffn_size=ffn).eval()
    with torch.no_grad():  # (biases and the position biases away from their zero / symmetric initial values)
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return net


@pytest.mark.parametrize("shape", [(2, 3, 12), (1, 2, 37), (1, 1, 4)], ids=lambda s: "x".join(map(str, s)))
def test_native_nbc_forward_fp32(backend, shape):
    from nbss_amd.nbc import NativeNBC
    B, F, T = shape
    net = _net(48, 2, 64)
    x = torch.randn(B, F, T, 4, generator=torch.Generator().manual_seed(5))
    want = net(x)
    got = NativeNBC(net.to(backend.device), backend.lib).forward(x.to(backend.device))
    assert got.shape == want.shape
    assert rel_l2(got, want) < 2e-5


def test_native_nbc_forward_head_width_48_bf16(backend):
    from nbss_amd.nbc import NativeNBC
    net = _net(96, 2, 128, layers=1)
    x = torch.randn(1, 2, 21, 4, generator=torch.Generator().manual_seed(6))
    want = net(x)
    got = NativeNBC(net.to(backend.device), backend.lib).forward(x.to(backend.device).to(torch.bfloat16))
    assert rel_l2(got.float(), want) < 3e-2


@pytest.mark.parametrize("dtype", [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")])
def test_relpos_attention_and_group_norm_blocks(backend, dtype):
    lib, dev = backend.lib, backend.device
    td = torch.bfloat16 if dtype == NBSS_BF16 else torch.float32
    tol = 2e-5 if dtype == NBSS_F32 else 1.5e-2
    g = torch.Generator().manual_seed(0)
    nseq, T, heads, dh = 3, 19, 2, 24
    H = heads * dh
    qkv = torch.randn(nseq, T, 3 * H, generator=g).to(td).to(dev)
    pos = torch.randn(2 * T - 1, H, generator=g).to(td).to(dev)
    u, v = (torch.randn(heads, dh, generator=g) * 0.5).to(dev), (torch.randn(heads, dh, generator=g) * 0.5).to(dev)
    scale = 1.0 / math.sqrt(H)
    o = torch.empty(nseq, T, H, dtype=td, device=dev)
    lib.call("nbss_nb_attention_relpos_fwd", dtype, nseq, T, H, heads, ops._ptr(lib, qkv), ops._ptr(lib, pos), ops._ptr(lib, u), ops._ptr(lib, v), scale,
             ops._ptr(lib, o), ops._stream(lib, qkv))
    q, k, vv = [t.double().cpu().view(nseq, T, heads, dh).transpose(1, 2) for t in qkv.split(H, -1)]
    P = pos.double().cpu().view(2 * T - 1, heads, dh).permute(1, 2, 0)
    content = (q + u.double().cpu()[None, :, None]) @ k.transpose(-1, -2)
    qp = (q + v.double().cpu()[None, :, None]) @ P
    idx = torch.arange(T)
    rel = (idx[:, None] - idx[None, :] + T - 1).expand(nseq, heads, T, T)
    want = (torch.softmax((content + qp.gather(-1, rel)) * scale, -1) @ vv).transpose(1, 2).reshape(nseq, T, H)
    assert rel_l2(o, want) < tol
    # GroupNorm(4, 64) over (16 channels x T) per sequence, with and without SiLU
    x = torch.randn(nseq, T, 64, generator=g).to(td).to(dev)
    gam, bet = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.1).to(dev)
    for act in (0, 1):
        y = torch.empty_like(x)
        lib.call("nbss_nb_group_norm", dtype, nseq, T, 64, 4, ops._ptr(lib, x), ops._ptr(lib, gam), ops._ptr(lib, bet), act, ops._ptr(lib, y), ops._stream(lib, x))
        ref = Fn.group_norm(x.double().cpu().transpose(1, 2), 4, gam.double().cpu(), bet.double().cpu(), 1e-5).transpose(1, 2)
        assert rel_l2(y, Fn.silu(ref) if act else ref) < tol


@pytest.mark.gpu
def test_nbc_module_takes_the_native_path_on_the_device(hip_lib, monkeypatch):
    """models.arch.NBC.NBC.forward (eval, no grad, HIP tensor: the default) = the torch.nn modules on the same device (NBSS_NBC_NATIVE=0); BASELINE-like
    widths (192 / 8 heads / 384) at 129 frequencies x 251 frames in, and a refusal of the kernels falls through to the modules with a warning"""
    import warnings

    import models.arch.NBC as M
    net = _net(192, 8, 384, layers=2, din=16, dout=4).cuda()
    x = torch.randn(1, 129, 251, 16, generator=torch.Generator().manual_seed(8)).cuda()
    with torch.no_grad():
        monkeypatch.setenv("NBSS_NBC_NATIVE", "0")
        with pytest.warns(RuntimeWarning, match="NBSS_NBC_NATIVE=0"):
            want = net(x)
        monkeypatch.delenv("NBSS_NBC_NATIVE")
        M._NATIVE.pop(net, None)
        assert net._native() is not None
        with warnings.catch_warnings():
            warnings.simplefilter("error")  # the native path is silent
            got = net(x)
        assert not torch.equal(got, want)  # (another implementation ran)
        assert rel_l2(got, want) < 1e-4
        # 300 frames: beyond the attention kernel's LDS table -> torch.nn modules, one warning
        x2 = torch.randn(1, 3, 300, 16, generator=torch.Generator().manual_seed(9)).cuda()
        with pytest.warns(RuntimeWarning, match="300 frames"):
            y2 = net(x2)
        assert y2.shape == (1, 3, 300, 4)
