"""SpatialNet-large geometry (dim_hidden 192, dim_ffn 384, dim_squeeze 16, 4 heads of 48; configs/SpatialNet.yaml "for large" comments) —
the forward (inference) kernels against the fp64 oracle, block by block and as a whole network, both stream dtypes; plus the generalised
T-ConvFFN tiling instantiated at the SMALL geometry against the same oracle (it is the same template).  Training entry points refuse."""
import pytest
import torch

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, NBSS_F32, NbssError
from oracle import spatialnet_ref as ref
from util import Case, rel_l2

DTYPES = [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")]


def shapes(backend, long_t=False):
    s = [(1, 5, 19), (2, 33, 40)]
    if backend.name == "hip":
        s += [(2, 129, 251), (1, 129, 600)] if long_t else [(2, 129, 251)]
    return s


@pytest.mark.parametrize("dtype", DTYPES)
def test_large_encoder_decoder(backend, dtype):
    for (B, F, T) in shapes(backend):
        cs = Case(backend, B, F, T, dtype, geo="large")
        xin, xin64 = cs.stream(seed=5, H=12)
        y = ops.encoder_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, xin)
        assert rel_l2(y, ref.encoder(xin64, cs.p64)) < cs.tol
        x, x64 = cs.stream(seed=6)
        o = ops.decoder_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, x)
        assert rel_l2(o, ref.decoder(x64, cs.p64)) < cs.tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("which", [0, 1])
def test_large_fconv(backend, dtype, which):
    for (B, F, T) in shapes(backend) + ([(1, 257, 3)] if dtype == NBSS_BF16 else []):  # (fp32 at 257 bins: 212 KB of LDS image, refused)
        cs = Case(backend, B, F, T, dtype, geo="large")
        x, x64 = cs.stream(seed=7)
        y = ops.fconv_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, which, x)
        want = ref.fconv(x64, cs.p64, f"layers.0.fconv{which + 1}")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_large_full(backend, dtype):
    for (B, F, T) in shapes(backend) + ([(1, 257, 3)] if dtype == NBSS_BF16 else []):
        cs = Case(backend, B, F, T, dtype, geo="large")
        x, x64 = cs.stream(seed=8)
        y = ops.full_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
        want = ref.full(x64, cs.p64, "layers.0")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_large_mhsa(backend, dtype):
    for (B, F, T) in shapes(backend, long_t=True) + [(1, 2, 300)]:  # 300 frames: three key blocks at 128 keys (fp32: five at 64)
        cs = Case(backend, B, F, T, dtype, geo="large")
        x, x64 = cs.stream(seed=9)
        o = ops.mhsa_save(cs.lib, cs.cfg, backend.device)
        y = ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=o)
        want = ref.mhsa(x64, cs.p64, "layers.0")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_large_tconvffn(backend, dtype):
    for (B, F, T) in shapes(backend, long_t=True) + [(1, 2, 300)]:  # 300 frames: several chunks with halos, sequence-wide GroupNorm
        cs = Case(backend, B, F, T, dtype, geo="large")
        x, x64 = cs.stream(seed=10)
        y = ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
        want = ref.tconvffn(x64, cs.p64, "layers.0")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_large_network_forward(backend, dtype):
    """whole network through nbss_spatialnet_fwd (12 layers at the reference's input shape on the GPU): fp32 stream <= 1e-3 (north-star
    bar), bf16 <= 3e-2; training mode is refused"""
    from nbss_amd.engine import SpatialNetEngine
    B, F, T, L = (1, 9, 21, 2) if backend.name == "emu" else (1, 129, 251, 12)
    kw = dict(dim_hidden=192, dim_ffn=384, dim_squeeze=16)
    p = ref.init_params(num_layers=L, num_freqs=F, dim_input=12, dim_output=4, seed=4, **kw)
    eng = SpatialNetEngine(backend.lib, backend.device, dim_input=12, dim_output=4, num_freqs=F, num_layers=L, dtype=dtype, **kw)
    eng.load_params(p)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, F, T, 12, generator=g).to(eng.stream_dtype())
    y = eng.forward(x.to(backend.device), train=False)
    want = ref.spatialnet(x.double(), {k: v.double() for k, v in p.items()}, L)
    assert rel_l2(y, want) < (1e-3 if dtype == NBSS_F32 else 3e-2)
    with pytest.raises(NbssError):
        eng.forward(x.to(backend.device), train=True)


def test_large_is_forward_only(backend):
    cs = Case(backend, 1, 5, 19, NBSS_BF16, geo="large")
    x, _ = cs.stream(seed=1)
    dy, _ = cs.stream(seed=2)
    G = torch.zeros_like(cs.flat)
    ws = ops.workspace(cs.lib, cs.cfg, backend.device)
    with pytest.raises(NbssError, match="UNSUPPORTED"):
        ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws)
    with pytest.raises(NbssError, match="UNSUPPORTED"):
        ops.fconv_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, 0, x, dy, ws)


@pytest.mark.gpu
def test_large_dropin_module_inference(hip_lib):
    """models.arch.SpatialNet.SpatialNet with the large configuration: eval-mode forward (what validate / test / predict run) equals the
    oracle on the module's own state_dict; a training-mode forward raises"""
    from models.arch.SpatialNet import SpatialNet
    torch.manual_seed(5)
    net = SpatialNet(dim_input=12, dim_output=4, num_layers=3, dim_hidden=192, dim_ffn=384, num_heads=4, dim_squeeze=16, num_freqs=129).cuda().eval()
    x = torch.randn(1, 129, 40, 12, device="cuda")
    with torch.no_grad():
        y = net(x)
    p = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
    assert rel_l2(y, ref.spatialnet(x.double().cpu(), p, 3)) < 1e-3
    net.train()
    with pytest.raises((NbssError, RuntimeError)):
        net(x).sum().backward()
