"""Native NBC2 forward (nbss_amd/nbc2.py over the nbss_nb_* building blocks of the C ABI) against the torch.nn module it reads its parameters
from (models/arch/NBC2.py, itself pinned to the reference's NBC2 by tests/test_nb_models.py), and each building block against torch."""
import pytest
import torch
import torch.nn.functional as Fn

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, NBSS_F32
from util import rel_l2

DTYPES = [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")]


def _td(dtype):
    return torch.bfloat16 if dtype == NBSS_BF16 else torch.float32


@pytest.mark.parametrize("dtype", DTYPES)
def test_building_blocks(backend, dtype):
    import ctypes as C
    lib, dev, td = backend.lib, backend.device, _td(dtype)
    tol = 2e-5 if dtype == NBSS_F32 else 1.5e-2
    g = torch.Generator().manual_seed(0)
    nseq, T = 6, 37

    def dv(t):
        return t.to(td).to(dev).contiguous()

    def conv(x, cin, ldx, cout, groups, taps, w, b, res=None, act_in=0, act_out=0):
        y = torch.empty(nseq, T, cout, dtype=td, device=dev)
        ws = torch.empty(lib._dll.nbss_nb_ws_bytes(cout, (cin + 7) // 8 * 8 if groups == 1 else cin, groups, taps), dtype=torch.uint8, device=dev)
        w, b = w.to(dev), b.to(dev)  # (named: a temporary's device memory is handed to the next allocation before the call that reads it is even made)
        lib.call("nbss_nb_conv_t", dtype, nseq, T, cin, ldx, cout, groups, taps, ops._ptr(lib, x), ops._ptr(lib, w), ops._ptr(lib, b), ops._ptr(lib, y),
                 ops._ptr(lib, res) if res is not None else None, act_in, act_out, ops._ptr(lib, ws), ops._stream(lib, x))
        return y

    # grouped conv along T with SiLU on the input and a residual
    x = torch.randn(nseq, T, 48, generator=g)
    w, b = torch.randn(48, 24, 3, generator=g) * 0.2, torch.randn(48, generator=g) * 0.1
    res = torch.randn(nseq, T, 48, generator=g)
    xd, rd = dv(x), dv(res)
    y = conv(xd, 48, 48, 48, 2, 3, w, b, res=rd, act_in=1)
    want = rd.double().cpu() + Fn.conv1d(Fn.silu(xd.double().cpu()).transpose(1, 2), w.double(), b.double(), padding="same", groups=2).transpose(1, 2)
    assert rel_l2(y, want) < tol
    # dense odd-width input (12 valid of 16 stored columns), kernel 5, SiLU on the output
    x = torch.zeros(nseq, T, 16)
    x[..., :12] = torch.randn(nseq, T, 12, generator=g)
    w, b = torch.randn(40, 12, 5, generator=g) * 0.2, torch.randn(40, generator=g) * 0.1
    xd = dv(x)
    y = conv(xd, 12, 16, 40, 1, 5, w, b, act_out=1)
    want = Fn.silu(Fn.conv1d(xd[..., :12].double().cpu().transpose(1, 2), w.double(), b.double(), padding="same").transpose(1, 2))
    assert rel_l2(y, want) < tol
    # LayerNorm
    x = torch.randn(nseq, T, 96, generator=g)
    gam, bet = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g) * 0.1
    xd = dv(x)
    y, stats = torch.empty_like(xd), torch.empty(nseq * T, 2, device=dev)
    gd, bd = gam.to(dev), bet.to(dev)
    lib.call("nbss_nb_layernorm", dtype, nseq * T, 96, ops._ptr(lib, xd), ops._ptr(lib, gd), ops._ptr(lib, bd), ops._ptr(lib, y), ops._ptr(lib, stats),
             ops._stream(lib, xd))
    assert rel_l2(y, Fn.layer_norm(xd.double().cpu(), (96,), gam.double(), bet.double(), 1e-5)) < tol
    # GroupBatchNorm: 2 utterances of 3 sequences, statistics per (utterance, frame) over (sequence, feature)
    from models.arch.NBC2 import GroupBatchNorm
    m = GroupBatchNorm(96, 3).double()
    with torch.no_grad():
        m.weight.copy_(gam.double())
        m.bias.copy_(bet.double())
    y = torch.empty_like(xd)
    lib.call("nbss_nb_group_batch_norm", dtype, 2, 3, T, 96, ops._ptr(lib, xd), ops._ptr(lib, gd), ops._ptr(lib, bd), C.c_float(1e-5), 1, ops._ptr(lib, y),
             ops._stream(lib, xd))
    assert rel_l2(y, Fn.silu(m(xd.double().cpu())).detach()) < tol
    # attention, 2 heads of 48
    qkv = torch.randn(nseq, T, 288, generator=g)
    qd = dv(qkv)
    o = torch.empty(nseq, T, 96, dtype=td, device=dev)
    lib.call("nbss_nb_attention_fwd", dtype, nseq, T, 96, 2, ops._ptr(lib, qd), ops._ptr(lib, o), ops._stream(lib, qd))
    q, k, v = [t.reshape(nseq, T, 2, 48).transpose(1, 2) for t in qd.double().cpu().split(96, dim=-1)]
    want = (torch.softmax(q @ k.transpose(-1, -2) / 48 ** 0.5, -1) @ v).transpose(1, 2).reshape(nseq, T, 96)
    assert rel_l2(o, want) < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_native_nbc2_forward_equals_the_module(backend, dtype):
    from models.arch.NBC2 import NBC2
    from nbss_amd.nbc2 import NativeNBC2, supported
    torch.manual_seed(3)
    B, F, T = (2, 5, 21) if backend.name == "emu" else (2, 129, 251)
    net = NBC2(dim_input=16, dim_output=6, n_layers=2 if backend.name == "emu" else 8, dim_hidden=96, dim_ffn=192, num_freqs=F).eval()
    assert supported(net) is None
    x = torch.randn(B, F, T, 16)
    with torch.no_grad():
        want = net.double()(x.double())
    net = net.float().to(backend.device)
    xd = x.to(_td(dtype)).to(backend.device)
    y = NativeNBC2(net, backend.lib).forward(xd)
    assert y.shape == (B, F, T, 6) and y.dtype == xd.dtype
    assert rel_l2(y, want) < (1e-4 if dtype == NBSS_F32 else 4e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_native_nbc2_training_gradients_equal_autograd(backend, dtype):
    """NativeNBC2.forward_train (one autograd.Function; backward over the nbss_nb_*_bwd building blocks): output and EVERY parameter gradient against
    torch.autograd through the torch.nn module in fp64 (the module is pinned to the reference's NBC2 by tests/test_nb_models.py) — 8 channels -> 3
    speakers (BASELINE config 4's interface), per-frame GroupBatchNorm over the utterance's frequencies"""
    from models.arch.NBC2 import NBC2
    from nbss_amd.nbc2 import NativeNBC2
    torch.manual_seed(5)
    B, F, T, L = (2, 5, 21, 2) if backend.name == "emu" else (1, 129, 251, 2)  # (the fp64 torch reference of the GPU case runs on the host: kept to ~20 s)
    net = NBC2(dim_input=16, dim_output=6, n_layers=L, dim_hidden=96, dim_ffn=192, num_freqs=F)
    with torch.no_grad():  # non-trivial norm affines / biases everywhere
        for p in net.parameters():
            if p.dim() <= 2 and p.shape[-1] in (1, 96, 192, 288, 6) and p.dim() == 1 or p.dim() == 2 and p.shape[-1] == 1:
                p.add_(0.1 * torch.randn_like(p))
    x = torch.randn(B, F, T, 16)
    r = torch.randn(B, F, T, 6)
    ref = NBC2(dim_input=16, dim_output=6, n_layers=L, dim_hidden=96, dim_ffn=192, num_freqs=F).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    xs = x.to(_td(dtype))
    (ref(xs.double()) * r.double()).sum().backward()  # (the reference sees the same rounded input as the stream)
    want_y = ref(xs.double()).detach()
    net = net.float().to(backend.device).train()
    y = NativeNBC2(net, backend.lib).forward_train(xs.to(backend.device))
    assert y.requires_grad and y.shape == (B, F, T, 6)
    (y.float() * r.to(backend.device)).sum().backward()
    assert rel_l2(y, want_y) < (1e-4 if dtype == NBSS_F32 else 4e-2)
    tol = 2e-4 if dtype == NBSS_F32 else 6e-2
    bad = {}
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        e = rel_l2(p.grad, q.grad)
        if e > tol:
            bad[n] = e
    assert not bad, bad


def test_supported_names_the_reason():
    from models.arch.NBC2 import NBC2
    from nbss_amd.nbc2 import supported
    assert supported(NBC2(dim_input=12, dim_output=4, n_layers=1, dim_hidden=96, dim_ffn=192, num_freqs=9)) is None
    bk = {"n_heads": 2, "dropout": 0, "conv_kernel_size": 3, "n_conv_groups": 8, "norms": ("LN", "LN", "GN")}
    assert "norms" in supported(NBC2(dim_input=12, dim_output=4, n_layers=1, dim_hidden=96, dim_ffn=192, num_freqs=9, block_kwargs=bk))
    bk = {"n_heads": 8, "dropout": 0, "conv_kernel_size": 3, "n_conv_groups": 8, "norms": ("LN", "GBN", "GBN"),
          "group_batch_norm_kwargs": {"share_along_sequence_dim": False}}
    assert "head width" in supported(NBC2(dim_input=12, dim_output=4, n_layers=1, dim_hidden=96, dim_ffn=192, num_freqs=9, block_kwargs=bk))


def test_conv_t_random_shapes(emu_lib):
    """the tap-GEMM behind every Linear / Conv1d of the generic paths against torch on random shapes (emulator): group counts, group widths that do and
    do not fill a 32-wide k-step or a 16-row tile, kernel sizes, sequence lengths around the 16-row tile and 64-row workgroup boundaries, both
    activation flags, residual on / off"""
    from hypothesis import given, settings, strategies as st
    lib, dev = emu_lib, torch.device("cpu")

    @settings(max_examples=60, deadline=None)
    @given(nseq=st.integers(1, 3), T=st.integers(1, 70), groups=st.sampled_from([1, 2, 8]), cgi=st.sampled_from([8, 24, 40]), cgo=st.sampled_from([4, 16, 24, 72]),
           taps=st.sampled_from([1, 3, 5]), act_in=st.integers(0, 1), act_out=st.integers(0, 1), res=st.booleans(), seed=st.integers(0, 1000))
    def check(nseq, T, groups, cgi, cgo, taps, act_in, act_out, res, seed):
        g = torch.Generator().manual_seed(seed)
        cin, cout = groups * cgi, groups * cgo
        x = torch.randn(nseq, T, cin, generator=g)
        w = torch.randn(cout, cgi, taps, generator=g) * 0.3
        b = torch.randn(cout, generator=g) * 0.1
        r = torch.randn(nseq, T, cout, generator=g) if res else None
        y = torch.empty(nseq, T, cout)
        ws = torch.empty(lib._dll.nbss_nb_ws_bytes(cout, cin, groups, taps), dtype=torch.uint8)
        wk = w.reshape(cout, cgi).contiguous() if taps == 1 else w
        lib.call("nbss_nb_conv_t", NBSS_F32, nseq, T, cin, cin, cout, groups, taps, ops._ptr(lib, x), ops._ptr(lib, wk), ops._ptr(lib, b), ops._ptr(lib, y),
                 ops._ptr(lib, r) if res else None, act_in, act_out, ops._ptr(lib, ws), None)
        xin = Fn.silu(x.double()) if act_in else x.double()
        want = Fn.conv1d(xin.transpose(1, 2), w.double(), b.double(), padding="same", groups=groups).transpose(1, 2)
        if act_out:
            want = Fn.silu(want)
        if res:
            want = want + r.double()
        assert rel_l2(y, want) < 2e-5, (nseq, T, groups, cgi, cgo, taps, act_in, act_out, res)

    check()


def test_attention_layernorm_gbn_random_shapes(emu_lib):
    """the attention, LayerNorm and GroupBatchNorm building blocks on random shapes (emulator): sequence lengths from 1 to 256 (partial 16-key tiles, odd
    tile pairs), both head widths, channel counts with and without the 16-lane row kernels"""
    import ctypes as C
    from hypothesis import given, settings, strategies as st
    from models.arch.NBC2 import GroupBatchNorm
    lib = emu_lib

    @settings(max_examples=20, deadline=None)
    @given(nseq=st.integers(1, 2), T=st.sampled_from([1, 2, 15, 16, 17, 31, 33, 48, 100, 255, 256]), dh=st.sampled_from([24, 48]), heads=st.integers(1, 3), seed=st.integers(0, 99))
    def attn(nseq, T, dh, heads, seed):
        g = torch.Generator().manual_seed(seed)
        H = dh * heads
        qkv = torch.randn(nseq, T, 3 * H, generator=g)
        o = torch.empty(nseq, T, H)
        lib.call("nbss_nb_attention_fwd", NBSS_F32, nseq, T, H, heads, ops._ptr(lib, qkv), ops._ptr(lib, o), None)
        q, k, v = [t.reshape(nseq, T, heads, dh).transpose(1, 2) for t in qkv.double().split(H, dim=-1)]
        want = (torch.softmax(q @ k.transpose(-1, -2) / dh ** 0.5, -1) @ v).transpose(1, 2).reshape(nseq, T, H)
        assert rel_l2(o, want) < 2e-5, (nseq, T, dh, heads)

    @settings(max_examples=20, deadline=None)
    @given(rows=st.integers(1, 70), Cc=st.sampled_from([8, 40, 96, 192, 384]), seed=st.integers(0, 99))
    def ln(rows, Cc, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(rows, Cc, generator=g) * 2 + 0.5
        gam, bet = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
        y, stats = torch.empty_like(x), torch.empty(rows, 2)
        lib.call("nbss_nb_layernorm", NBSS_F32, rows, Cc, ops._ptr(lib, x), ops._ptr(lib, gam), ops._ptr(lib, bet), ops._ptr(lib, y), ops._ptr(lib, stats), None)
        assert rel_l2(y, Fn.layer_norm(x.double(), (Cc,), gam.double(), bet.double(), 1e-5)) < 2e-5, (rows, Cc)

    @settings(max_examples=15, deadline=None)
    @given(B=st.integers(1, 2), F=st.integers(1, 5), T=st.integers(1, 9), Cc=st.sampled_from([8, 96, 192]), act=st.integers(0, 1), seed=st.integers(0, 99))
    def gbn(B, F, T, Cc, act, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B * F, T, Cc, generator=g) * 2 + 0.3
        m = GroupBatchNorm(Cc, F).double()
        with torch.no_grad():
            m.weight.copy_(torch.rand(Cc, generator=g).double() + 0.5)
            m.bias.copy_(torch.randn(Cc, generator=g).double())
        y = torch.empty_like(x)
        gw, gb = m.weight.float().contiguous(), m.bias.float().contiguous()
        lib.call("nbss_nb_group_batch_norm", NBSS_F32, B, F, T, Cc, ops._ptr(lib, x), ops._ptr(lib, gw), ops._ptr(lib, gb), C.c_float(1e-5), act, ops._ptr(lib, y), None)
        want = m(x.double()).detach()
        assert rel_l2(y, Fn.silu(want) if act else want) < 2e-5, (B, F, T, Cc, act)

    attn()
    ln()
    gbn()


def test_training_blocks_random_shapes(emu_lib):
    """the backward building blocks (nbss_nb_conv_t_bwd with and without the SiLU' factor, group_batch_norm_bwd, layernorm_bwd, attention_bwd) against torch
    autograd on random shapes (emulator, fp32): group counts / widths, taps, padded dense inputs, partial tiles of the sequence axis"""
    import ctypes as C
    from hypothesis import given, settings, strategies as st
    from models.arch.NBC2 import GroupBatchNorm
    lib = emu_lib
    P = lambda t: ops._ptr(lib, t)  # noqa: E731

    @settings(max_examples=25, deadline=None)
    @given(nseq=st.integers(1, 3), T=st.sampled_from([1, 2, 7, 16, 33]), groups=st.sampled_from([1, 1, 2, 4]), cgi=st.sampled_from([8, 16, 24]), cgo=st.sampled_from([8, 24, 40]),
           taps=st.sampled_from([1, 3, 5]), pre=st.booleans(), pad=st.booleans(), seed=st.integers(0, 1000))
    def conv(nseq, T, groups, cgi, cgo, taps, pre, pad, seed):
        g = torch.Generator().manual_seed(seed)
        cin, cout = groups * cgi, groups * cgo
        valid = cin - 4 if (pad and groups == 1) else cin  # dense: the last stored columns may be padding (zero in x, no gradient wanted there)
        a = torch.randn(nseq, T, cin, generator=g)           # pre-activation of the input
        x = (Fn.silu(a) if pre else a.clone())
        x[..., valid:] = 0
        w = torch.randn(cout, valid // groups, taps, generator=g) * 0.3
        dy = torch.randn(nseq, T, cout, generator=g)
        dx = torch.full((nseq, T, cin), float("nan"))
        dw, db = torch.zeros(w.numel()), torch.zeros(cout)
        ws = torch.empty(lib._dll.nbss_nb_bwd_ws_bytes(cout, cin, groups, taps), dtype=torch.uint8)
        wk = w.reshape(cout, -1).contiguous() if taps == 1 else w
        lib.call("nbss_nb_conv_t_bwd", NBSS_F32, nseq, T, valid, cin, cout, groups, taps, P(x), P(wk), P(dy), P(a) if pre else None, P(dx), P(dw), P(db), P(ws), None)
        a64 = a.double().requires_grad_(True)
        w64 = w.double().requires_grad_(True)
        xin = (Fn.silu(a64) if pre else a64)[..., :valid]
        y = Fn.conv1d(xin.transpose(1, 2), w64, None, padding="same", groups=groups).transpose(1, 2)
        (y * dy.double()).sum().backward()
        assert rel_l2(dx[..., :valid], a64.grad[..., :valid]) < 2e-5, "dx"
        assert rel_l2(dw.reshape(w.shape), w64.grad) < 2e-5, "dw"
        assert rel_l2(db, dy.double().sum((0, 1))) < 2e-5, "db"

    @settings(max_examples=15, deadline=None)
    @given(B=st.integers(1, 2), F=st.integers(1, 5), T=st.integers(1, 9), Cc=st.sampled_from([8, 96, 192, 384]), act=st.integers(0, 1), seed=st.integers(0, 99))
    def gbn(B, F, T, Cc, act, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B * F, T, Cc, generator=g) * 2 + 0.3
        dy = torch.randn(B * F, T, Cc, generator=g)
        m = GroupBatchNorm(Cc, F).double()
        with torch.no_grad():
            m.weight.copy_(torch.rand(Cc, generator=g).double() + 0.5)
            m.bias.copy_(torch.randn(Cc, generator=g).double())
        gw, gb = m.weight.detach().float().contiguous(), m.bias.detach().float().contiguous()
        dx, dg, dbt = torch.empty_like(x), torch.zeros(Cc), torch.zeros(Cc)
        lib.call("nbss_nb_group_batch_norm_bwd", NBSS_F32, B, F, T, Cc, P(x), P(gw), P(gb), C.c_float(1e-5), act, P(dy), P(dx), P(dg), P(dbt), None)
        x64 = x.double().requires_grad_(True)
        y = m(x64)
        ((Fn.silu(y) if act else y) * dy.double()).sum().backward()
        assert rel_l2(dx, x64.grad) < 5e-5 and rel_l2(dg, m.weight.grad) < 5e-5 and rel_l2(dbt, m.bias.grad) < 5e-5, (B, F, T, Cc, act)

    @settings(max_examples=12, deadline=None)
    @given(nseq=st.integers(1, 2), T=st.sampled_from([1, 2, 15, 16, 17, 33, 100, 256]), dh=st.sampled_from([24, 48]), heads=st.integers(1, 2), seed=st.integers(0, 99))
    def attn(nseq, T, dh, heads, seed):
        g = torch.Generator().manual_seed(seed)
        H = dh * heads
        qkv = torch.randn(nseq, T, 3 * H, generator=g)
        do = torch.randn(nseq, T, H, generator=g)
        dqkv = torch.empty_like(qkv)
        ws = torch.empty(lib._dll.nbss_nb_attention_bwd_ws_bytes(NBSS_F32, nseq, T, H, heads), dtype=torch.uint8)
        lib.call("nbss_nb_attention_bwd", NBSS_F32, nseq, T, H, heads, P(qkv), P(do), P(dqkv), P(ws), None)
        q64 = qkv.double().requires_grad_(True)
        q, k, v = [t.reshape(nseq, T, heads, dh).transpose(1, 2) for t in q64.split(H, dim=-1)]
        o = (torch.softmax(q @ k.transpose(-1, -2) / dh ** 0.5, -1) @ v).transpose(1, 2).reshape(nseq, T, H)
        (o * do.double()).sum().backward()
        assert rel_l2(dqkv, q64.grad) < 5e-5, (nseq, T, dh, heads)

    conv()
    gbn()
    attn()


@pytest.mark.gpu
def test_module_dispatch_on_the_device(hip_lib):
    """models.arch.NBC2.NBC2.forward on a HIP tensor: the default norms take the native path silently; other norm types (no GroupBatchNorm, hence no
    `group_size`), a frequency count other than the group size and sequences beyond 256 frames run the torch.nn modules with ONE warning naming the reason
    (round 4 raised AttributeError for the first of these)"""
    import warnings

    from models.arch.NBC2 import NBC2
    torch.manual_seed(11)
    x = torch.randn(1, 9, 40, 12).cuda()
    net = NBC2(dim_input=12, dim_output=4, n_layers=1, dim_hidden=96, dim_ffn=192, num_freqs=9).cuda().eval()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        y = net(x)
    assert y.shape == (1, 9, 40, 4)
    bk = {"n_heads": 2, "dropout": 0, "conv_kernel_size": 3, "n_conv_groups": 8, "norms": ("LN", "LN", "GN")}
    other = NBC2(dim_input=12, dim_output=4, n_layers=1, dim_hidden=96, dim_ffn=192, num_freqs=9, block_kwargs=bk).cuda()
    for train in (False, True):
        other.train(train)
        with pytest.warns(RuntimeWarning, match="norms must be") if not train else warnings.catch_warnings():  # (one warning per module and reason)
            y = other(x)
        assert y.shape == (1, 9, 40, 4)
    with torch.no_grad():
        with pytest.warns(RuntimeWarning, match="group_size"):  # (18 sequences = two groups of 9 for the module; the kernels take whole utterances)
            assert net(torch.randn(1, 18, 40, 12).cuda()).shape == (1, 18, 40, 4)
        with pytest.warns(RuntimeWarning, match="300 frames"):
            assert net(torch.randn(1, 9, 300, 12).cuda()).shape == (1, 9, 300, 4)
