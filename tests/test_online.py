"""OnlineSpatialNet (BASELINE config 5): the drop-in module against fixtures produced by the reference's own
models/arch/OnlineSpatialNet.py (tests/golden/make_golden.py: online_models), the causality property the reference checks in its
__main__ (OnlineSpatialNet.py:422-426), the equivalence of the three retention evaluation orders (retention.py:303-326), the chunked
streaming interface against the whole-utterance forward, and — on the GPU — the HIP-graph replay of a streaming step."""
from pathlib import Path

import numpy as np
import pytest
import torch

from util import rel_l2

Z = np.load(Path(__file__).resolve().parent / "golden" / "online_tiny.npz")
KW = dict(dim_input=4, dim_output=4, num_layers=2, dim_squeeze=8, num_freqs=9, encoder_kernel_size=5, dim_hidden=32, dim_ffn=64, num_heads=4,
          dropout=(0, 0, 0), kernel_size=(5, 3), conv_groups=(8, 8), norms=["LN", "LN", "GN", "LN", "LN", "LN"], full_share=0)
VARIANTS = {"mhsa31": dict(attention="mhsa(31)"), "mhsa7": dict(attention="mhsa(7)"), "ret2": dict(attention="ret(2)", decay=[4, 5, 9, 10], rope=False)}


def _net(name, seed=11):
    from models.arch.OnlineSpatialNet import OnlineSpatialNet
    torch.manual_seed(seed)
    return OnlineSpatialNet(**KW, **VARIANTS[name]).eval()


@pytest.mark.parametrize("name", ["mhsa31", "ret2"])
def test_matches_reference_fixture(name):
    net = _net(name)
    sd = {k[len(name) + 7:]: torch.from_numpy(Z[k]) for k in Z.files if k.startswith(f"{name}/param/")}
    assert set(sd) == set(net.state_dict())  # same keys as the reference module (incl. the shared `full` tensors, pos.angle / pos.decay)
    net.load_state_dict(sd, strict=True)
    x, r = torch.from_numpy(Z[f"{name}/x"]), torch.from_numpy(Z[f"{name}/r"])
    y = net(x)
    assert rel_l2(y, torch.from_numpy(Z[f"{name}/y"])) < 2e-5
    (y * r).sum().backward()
    want = {k[len(name) + 6:]: torch.from_numpy(Z[k]) for k in Z.files if k.startswith(f"{name}/grad/")}
    got = dict(net.named_parameters())
    top = max(float(g.norm()) for g in want.values())
    for k, g in want.items():
        assert float((got[k].grad - g).norm()) <= 5e-4 * float(g.norm()) + 1e-6 * top, k


@pytest.mark.parametrize("name", ["mhsa7", "ret2"])
def test_causal(name):
    """frames 0..n-1 of the output do not depend on later input frames"""
    net = _net(name)
    x = torch.randn(1, 9, 40, 4)
    with torch.no_grad():
        assert float((net(x)[:, :, :25] - net(x[:, :, :25])).abs().max()) < 1e-5


def test_attention_window_is_applied():
    """'mhsa(N)': frame t attends to frames t-N+1..t only (what bounds the streaming state).  (The reference's own forward loses the
    band on torch >= 2: nn.MultiheadAttention's is_causal fast path ignores attn_mask when need_weights is False; see make_golden.py.)"""
    net = _net("mhsa7")
    x = torch.randn(1, 9, 30, 4)
    x2 = x.clone()
    x2[:, :, :10] += 1.0  # frames more than 7 + (5-1) + 3*2 conv taps behind the probe frame cannot matter in a 2-layer net ... choose far away
    with torch.no_grad():
        m = net.get_causal_mask(slen=12, device=x.device)
        assert torch.isinf(m[10, 3]) and m[10, 4] == 0 and m[10, 10] == 0 and torch.isinf(m[10, 11])


def test_retention_evaluation_orders_agree():
    from models.arch.base.retention import MultiScaleRetention, RetNetRelPos
    torch.manual_seed(0)
    E, H, T = 96, 4, 70
    pos, m = RetNetRelPos(E, H, recurrent_chunk_size=16, decay=[4, 5, 9, 10]), MultiScaleRetention(E, H, value_factor=2, share_qk=True)
    x = torch.randn(3, T, E)
    with torch.no_grad():
        y = m(x, pos(T), rope=False)
        yc = m(x, pos(T, chunkwise_recurrent=True), chunkwise_recurrent=True, rope=False)
        st, ys = {}, []
        for t in range(T):
            ys.append(m(x[:, [t]], pos(t + 1, activate_recurrent=True), incremental_state=st, rope=False))
    # the orders differ by positive per-frame scalings that the per-head RMS normalisation (eps 1e-6) removes
    assert rel_l2(yc, y) < 2e-3 and rel_l2(torch.cat(ys, 1), y) < 2e-3


@pytest.mark.parametrize("name,chunk", [("mhsa7", 4), ("mhsa7", 16), ("ret2", 8)])
def test_streaming_equals_whole_utterance(name, chunk):
    from models.arch.OnlineSpatialNet import OnlineStreamer
    net = _net(name)
    x = torch.randn(2, 9, 48, 4)
    with torch.no_grad():
        y = net(x)
        st = net.init_stream(2)
        ys = torch.cat([net.forward_stream(x[:, :, c:c + chunk], st) for c in range(0, 48, chunk)], 2)
        s = OnlineStreamer(net, 2, chunk, use_graph=False)
        yz = torch.cat([s.step(x[:, :, c:c + chunk]) for c in range(0, 48, chunk)], 2)
    tol = 1e-5 if name == "mhsa7" else 2e-3
    assert rel_l2(ys, y) < tol and rel_l2(yz, y) < tol


def test_mamba_raises():
    from models.arch.OnlineSpatialNet import OnlineSpatialNet
    with pytest.raises(NotImplementedError, match="mamba"):
        OnlineSpatialNet(**KW, attention="mamba(16,4)")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mhsa7", "ret2"])
def test_hipgraph_streaming_step(name):
    """the streaming step captured in a HIP graph and replayed == eager whole-utterance forward (BASELINE config 5)"""
    from models.arch.OnlineSpatialNet import OnlineStreamer
    dev = torch.device("cuda:0")
    net = _net(name).to(dev)
    x = torch.randn(2, 9, 64, 4, device=dev)
    with torch.no_grad():
        y = net(x)
    s = OnlineStreamer(net, 2, 8, device=dev, use_graph=True)
    ys = torch.cat([s.step(x[:, :, c:c + 8]) for c in range(0, 64, 8)], 2)
    assert s.graph is not None
    assert rel_l2(ys, y) < (1e-4 if name == "mhsa7" else 3e-3)


# ---- native streaming step (csrc/online.hip + the cross-band kernels): emulator on CPU, libnbss_hip.so + HIP graph with -m gpu -----------
NATIVE_KW = dict(dim_input=4, dim_output=4, num_layers=2, dim_squeeze=8, num_freqs=9, encoder_kernel_size=5, dim_hidden=96, dim_ffn=192, num_heads=4,
                 dropout=(0, 0, 0), kernel_size=(5, 3), conv_groups=(8, 8), norms=["LN", "LN", "GN", "LN", "LN", "LN"], full_share=0, attention="ret(2)",
                 decay=[4, 5, 9, 10], rope=False)


def _native_net(seed=5, **over):
    from models.arch.OnlineSpatialNet import OnlineSpatialNet
    torch.manual_seed(seed)
    net = OnlineSpatialNet(**{**NATIVE_KW, **over}).eval()
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return net


@pytest.mark.parametrize("chunk,over", [(8, {}), (5, {}), (16, {"attention": "ret(2,not_share_qk)"}), (8, {"attention": "mhsa(11)"}), (4, {"attention": "mhsa(40)"})])
def test_native_streaming_step_matches_module(backend, chunk, over):
    """the HIP streaming step (encoder / retention / T-ConvFFN kernels of csrc/online.hip + the cross-band kernels), chunk by chunk, against the
    module's own whole-utterance forward (parallel retention) and its torch streaming step (the module itself is pinned to the reference by
    tests/golden/online_tiny.npz above); fp32: <= 2e-3 against the parallel form (the orders differ by the per-head RMS eps), <= 2e-4 against
    the torch recurrent step"""
    from nbss_amd.online import NativeOnlineStreamer
    net = _native_net(**over).to(backend.device)
    T = 3 * chunk
    x = torch.randn(2, 9, T, 4, device=backend.device)
    with torch.no_grad():
        y = net(x)
        st = net.init_stream(2, device=backend.device)
        ys = torch.cat([net.forward_stream(x[:, :, c:c + chunk], st) for c in range(0, T, chunk)], 2)
    s = NativeOnlineStreamer(net, 2, chunk, device=backend.device, lib=backend.lib, use_graph=backend.name == "hip")
    yn = torch.cat([s.step(x[:, :, c:c + chunk]) for c in range(0, T, chunk)], 2)
    assert rel_l2(yn, ys) < 2e-4 and rel_l2(yn, y) < 2e-3, (rel_l2(yn, ys), rel_l2(yn, y))
    assert (s.graph is not None) == (backend.name == "hip")
    s.reset()  # a new utterance through the same (captured) step
    yn2 = torch.cat([s.step(x[:, :, c:c + chunk]) for c in range(0, T, chunk)], 2)
    assert torch.equal(yn, yn2)  # bitwise: the GroupNorm sums are folded in a fixed order (no atomics in the inference path)


@pytest.mark.parametrize("name,chunk", [("ret2", 8), ("mhsa31", 6)])
def test_native_streaming_step_matches_reference_fixture(backend, name, chunk):
    """one hop from the reference: tests/golden/online_w96.npz holds the REFERENCE module's whole-utterance output at the width the native step
    serves (dim_hidden 96, made by tests/golden/make_golden.py online96); the HIP streaming step, fed chunk by chunk with the reference's
    state_dict, reproduces it (fp32; the recurrent / windowed forms differ from the reference's parallel form by rounding only)"""
    from nbss_amd.online import NativeOnlineStreamer
    from models.arch.OnlineSpatialNet import OnlineSpatialNet
    Z96 = np.load(Path(__file__).resolve().parent / "golden" / "online_w96.npz")
    kw = dict(attention="ret(2)", decay=[4, 5, 9, 10], rope=False) if name == "ret2" else dict(attention="mhsa(31)")
    net = OnlineSpatialNet(**{**NATIVE_KW, **kw}).eval()
    sd = {k[len(name) + 7:]: torch.from_numpy(Z96[k]) for k in Z96.files if k.startswith(f"{name}/param/")}
    assert set(sd) == set(net.state_dict())
    net.load_state_dict(sd, strict=True)
    net = net.to(backend.device)
    x, want = torch.from_numpy(Z96[f"{name}/x"]).to(backend.device), torch.from_numpy(Z96[f"{name}/y"])
    s = NativeOnlineStreamer(net, 2, chunk, device=backend.device, lib=backend.lib, use_graph=backend.name == "hip")
    yn = torch.cat([s.step(x[:, :, c:c + chunk].contiguous()) for c in range(0, x.shape[2], chunk)], 2)
    assert rel_l2(yn, want) < 2e-3, rel_l2(yn, want)


@pytest.mark.gpu
@pytest.mark.parametrize("attention", ["ret(2)", "mhsa(251)"])
def test_native_streaming_step_at_the_config5_geometry(hip_lib, attention):
    """BASELINE config 5's geometry on the GPU: 129 frequencies, 8 layers, 6 channels -> 2 speakers, 2 000 frames (32 s) in 16-frame chunks, one HIP
    graph per chunk — the native step against the torch.nn streaming step of the same module (<= 1e-4), and bitwise equal to itself on a second pass
    over the same utterance (no float atomics in the inference path)"""
    from models.arch.OnlineSpatialNet import OnlineSpatialNet, OnlineStreamer
    from nbss_amd.online import NativeOnlineStreamer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = OnlineSpatialNet(dim_input=12, dim_output=4, num_layers=8, dim_squeeze=8, num_freqs=129, encoder_kernel_size=5, dim_hidden=96, dim_ffn=192,
                           num_heads=4, dropout=(0, 0, 0), kernel_size=(5, 3), conv_groups=(8, 8), norms=["LN", "LN", "GN", "LN", "LN", "LN"], full_share=0,
                           attention=attention, decay=[4, 5, 9, 10], rope=False).eval().to(dev)
    chunk, T = 16, 2000
    x = torch.randn(1, 129, T, 12, device=dev)
    s = NativeOnlineStreamer(net, 1, chunk, device=dev, lib=hip_lib, use_graph=True)
    yn = torch.cat([s.step(x[:, :, c:c + chunk]) for c in range(0, T, chunk)], 2)
    assert s.graph is not None
    s.reset()
    yn2 = torch.cat([s.step(x[:, :, c:c + chunk]) for c in range(0, T, chunk)], 2)
    assert torch.equal(yn, yn2)
    t = OnlineStreamer(net, 1, chunk, device=dev, use_graph=True)
    yt = torch.cat([t.step(x[:, :, c:c + chunk]) for c in range(0, T, chunk)], 2)
    assert torch.isfinite(yn).all() and rel_l2(yn, yt) < 1e-4, rel_l2(yn, yt)


def test_native_streaming_refuses_other_geometries():
    from nbss_amd.online import NativeOnlineStreamer, supported
    assert supported(_native_net()) is None
    assert supported(_native_net(attention="mhsa(7)")) is None
    for over in ({"attention": "mhsa(inf)"}, {"dim_hidden": 32, "dim_ffn": 64}, {"rope": True}, {"attention": "mhsa(7)", "rope": "ALiBi"}):
        net = _native_net(**over)
        assert supported(net) is not None
        with pytest.raises(NotImplementedError):
            NativeOnlineStreamer(net, 1, 8, device="cpu", lib=object())
