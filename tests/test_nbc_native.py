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


def _pack_keep_bits(keep: torch.Tensor) -> torch.Tensor:
    """bool [nseq, heads, T, T] -> int32 words [nseq, heads, T, ceil(T / 32)]: bit (j & 31) of word j >> 5 = (i, j) kept"""
    nseq, heads, T, _ = keep.shape
    MW = (T + 31) // 32
    pad = torch.zeros(nseq, heads, T, MW * 32, dtype=torch.int64)
    pad[..., :T] = keep.to(torch.int64)
    words = (pad.view(nseq, heads, T, MW, 32) << torch.arange(32, dtype=torch.int64)).sum(-1)
    return (words - ((words >> 31) << 32)).to(torch.int32)  # two's complement int32 with the same bits


def _relpos_reference(qkv, pos, u, v, scale, keep, keep_scale, heads):
    nseq, T, H3 = qkv.shape
    H, dh = H3 // 3, H3 // 3 // heads
    q, k, vv = [t.view(nseq, T, heads, dh).transpose(1, 2) for t in qkv.split(H, -1)]
    P = pos.view(2 * T - 1, heads, dh).permute(1, 2, 0)
    content = (q + u[None, :, None]) @ k.transpose(-1, -2)
    qp = (q + v[None, :, None]) @ P
    idx = torch.arange(T)
    rel = (idx[:, None] - idx[None, :] + T - 1).expand(nseq, heads, T, T)
    p = torch.softmax((content + qp.gather(-1, rel)) * scale, -1)
    if keep is not None:
        p = p * keep.to(p.dtype) * keep_scale
    return (p @ vv).transpose(1, 2).reshape(nseq, T, H)


@pytest.mark.parametrize("dtype", [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")])
@pytest.mark.parametrize("case", [(3, 19, 2, 24, True), (2, 40, 2, 24, False), (1, 33, 1, 48, True)], ids=lambda c: f"n{c[0]}-T{c[1]}-h{c[2]}x{c[3]}-{'drop' if c[4] else 'nodrop'}")
def test_relpos_attention_backward(backend, dtype, case):
    """nbss_nb_attention_relpos_train / _bwd (three kernels: per query tile, per key tile, per offset block) against torch.autograd through the reference
    formula in fp64 with the SAME dropout keep-bits: output, dqkv, the position table's gradient (summed over the sequences) and the two bias gradients"""
    lib, dev = backend.lib, backend.device
    nseq, T, heads, dh, drop = case
    if backend.name == "hip":
        nseq, T = (6, 251) if dh == 24 else (3, 100)  # (head width 48 in fp32: the key-tile kernel holds three [T][48] images beside P in LDS)
    td = torch.bfloat16 if dtype == NBSS_BF16 else torch.float32
    tol = 5e-5 if dtype == NBSS_F32 else 3e-2
    g = torch.Generator().manual_seed(1)
    H = heads * dh
    qkv = torch.randn(nseq, T, 3 * H, generator=g).to(td)
    pos = torch.randn(2 * T - 1, H, generator=g).to(td)
    u, v = torch.randn(heads, dh, generator=g) * 0.5, torch.randn(heads, dh, generator=g) * 0.5
    dO = torch.randn(nseq, T, H, generator=g).to(td)
    scale = 1.0 / math.sqrt(H)
    keep = (torch.rand(nseq, heads, T, T, generator=g) > 0.25) if drop else None
    ks = 1.0 / 0.75
    q64, p64, u64, v64 = [t.double().requires_grad_(True) for t in (qkv, pos, u, v)]
    want_o = _relpos_reference(q64, p64, u64, v64, scale, keep, ks, heads)
    (want_o * dO.double()).sum().backward()
    bits = _pack_keep_bits(keep).to(dev) if drop else None
    qd, pd, ud, vd, dod = qkv.to(dev), pos.to(dev), u.to(dev).contiguous(), v.to(dev).contiguous(), dO.to(dev)
    o = torch.empty_like(qd[..., :H]).contiguous()
    st = ops._stream(lib, qd)
    P = lambda t: ops._ptr(lib, t)  # noqa: E731
    lib.call("nbss_nb_attention_relpos_train", dtype, nseq, T, H, heads, P(qd), P(pd), P(ud), P(vd), scale, P(bits), ks, P(o), st)
    assert rel_l2(o, want_o.detach()) < tol
    ws = torch.empty(lib._dll.nbss_nb_attention_relpos_bwd_ws_bytes(nseq, T, H, heads), dtype=torch.uint8, device=dev)
    dqkv = torch.empty_like(qd)
    dpos = torch.zeros(2 * T - 1, H, dtype=torch.float32, device=dev)
    du, dv = torch.zeros(H, dtype=torch.float32, device=dev), torch.zeros(H, dtype=torch.float32, device=dev)
    lib.call("nbss_nb_attention_relpos_bwd", dtype, nseq, T, H, heads, P(qd), P(pd), P(ud), P(vd), scale, P(bits), ks, P(dod), P(dqkv), P(dpos), P(du), P(dv), P(ws), st)
    for name, got, want in (("dq", dqkv[..., :H], q64.grad[..., :H]), ("dk", dqkv[..., H:2 * H], q64.grad[..., H:2 * H]), ("dv", dqkv[..., 2 * H:], q64.grad[..., 2 * H:]),
                            ("dpos", dpos, p64.grad), ("du", du.view(heads, dh), u64.grad), ("dvb", dv.view(heads, dh), v64.grad)):
        assert rel_l2(got, want) < tol, (name, rel_l2(got, want))


@pytest.mark.parametrize("dtype", [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")])
def test_group_norm_train_and_backward(backend, dtype):
    lib, dev = backend.lib, backend.device
    td = torch.bfloat16 if dtype == NBSS_BF16 else torch.float32
    tol = 5e-5 if dtype == NBSS_F32 else 3e-2
    g = torch.Generator().manual_seed(2)
    nseq, T, C, G = (4, 23, 96, 2) if backend.name != "hip" else (20, 248, 384, 8)
    x = torch.randn(nseq, T, C, generator=g).to(td)
    dy = torch.randn(nseq, T, C, generator=g).to(td)
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    x64, g64, b64 = [t.double().requires_grad_(True) for t in (x, gam, bet)]
    from models.arch.base.norm import group_norm
    want = Fn.silu(group_norm(x64.transpose(1, 2), G, g64, b64, 1e-5).transpose(1, 2))
    (want * dy.double()).sum().backward()
    xd, gd, bd = x.to(dev), gam.to(dev), bet.to(dev)
    y = torch.empty_like(xd)
    stats = torch.empty(nseq * G, 2, dtype=torch.float32, device=dev)
    P = lambda t: ops._ptr(lib, t)  # noqa: E731
    st = ops._stream(lib, xd)
    lib.call("nbss_nb_group_norm_train", dtype, nseq, T, C, G, P(xd), P(gd), P(bd), 1, P(y), P(stats), st)
    assert rel_l2(y, want.detach()) < tol
    dx = dy.to(dev).clone()
    dg, db = torch.zeros(C, dtype=torch.float32, device=dev), torch.zeros(C, dtype=torch.float32, device=dev)
    lib.call("nbss_nb_group_norm_bwd", dtype, nseq, T, C, G, P(xd), P(stats), P(gd), P(bd), P(dx), P(dg), P(db), st)
    assert rel_l2(dx, x64.grad) < tol and rel_l2(dg, g64.grad) < tol and rel_l2(db, b64.grad) < tol


def _zero_dropout(net):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


@pytest.mark.parametrize("dtype", [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")])
def test_native_nbc_training_gradients_equal_autograd(backend, dtype):
    """NativeNBC.forward_train (one autograd.Function; backward over the nbss_nb_*_bwd building blocks + the relative-position attention backward): output and
    EVERY parameter gradient against torch.autograd through the torch.nn module in fp64 (pinned to the reference's NBC by tests/test_nb_models.py), with the
    dropouts at p = 0 (the masks are random: the attention dropout's arithmetic is checked with given keep-bits in test_relpos_attention_backward)"""
    from nbss_amd.nbc import NativeNBC, train_supported
    hip = backend.name == "hip"
    B, F, T = (1, 129, 251) if hip else (2, 3, 23)
    hidden, heads, ffn, L = (192, 8, 384, 2) if hip else (48, 2, 64, 2)
    net = _zero_dropout(_net(hidden, heads, ffn, layers=L, din=16 if hip else 4, dout=4)).train()
    assert train_supported(net) is None
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, F, T, 16 if hip else 4, generator=g)
    r = torch.randn(B, F, T, 4, generator=g)
    td = torch.bfloat16 if dtype == NBSS_BF16 else torch.float32
    xs = x.to(td)
    import copy
    ref = copy.deepcopy(net).double()
    want_y = ref(xs.double())
    (want_y * r.double()).sum().backward()
    net = net.float().to(backend.device)
    y = NativeNBC(net, backend.lib).forward_train(xs.to(backend.device))
    assert y.requires_grad and y.shape == want_y.shape
    (y.float() * r.to(backend.device)).sum().backward()
    assert rel_l2(y, want_y.detach()) < (1e-4 if dtype == NBSS_F32 else 4e-2)
    tol = 3e-4 if dtype == NBSS_F32 else 6e-2
    bad = {}
    top = max(float(q.grad.norm()) for q in ref.parameters())
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        err = float((p.grad.double().cpu() - q.grad).norm())
        if err > tol * float(q.grad.norm()) + (1e-6 if dtype == NBSS_F32 else 1e-3) * top:  # (the key bias is analytically gradient-free under the softmax: absolute floor)
            bad[n] = (err, float(q.grad.norm()))
    assert not bad, bad


def test_native_nbc_training_with_dropout_is_seeded(backend):
    """the reference's NBC always trains with dropout 0.1: the native path draws its masks (attention keep-bits + three element-wise masks per block) from
    torch's generator on the device — the same seed gives the same loss and gradients, another seed other ones, eval mode none"""
    from nbss_amd.nbc import NativeNBC
    net = _net(48, 2, 64, layers=1).train().to(backend.device)
    x = torch.randn(1, 3, 19, 4, generator=torch.Generator().manual_seed(1)).to(backend.device)
    run = NativeNBC(net, backend.lib)

    def once(seed):
        torch.manual_seed(seed)
        net.zero_grad()
        y = run.forward_train(x)
        y.square().sum().backward()
        return y.detach().clone(), net.encoder.weight.grad.clone()

    y0, g0 = once(3)
    y1, g1 = once(3)
    y2, g2 = once(4)
    assert torch.isfinite(y0).all() and torch.isfinite(g0).all()
    assert torch.equal(y0, y1) and torch.allclose(g0, g1, rtol=1e-4, atol=1e-6)
    assert not torch.allclose(y0, y2)


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
    # training mode with autograd: the native training path (silent), gradients on every parameter
    net.train()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        y3 = net(x[:, :9, :64])
    assert y3.requires_grad and type(y3.grad_fn).__name__ == "_NBCTrainFnBackward"
    y3.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
