"""Forward parity of every SpatialNet sub-block kernel against the fp64 oracle (same seeded inputs).
Runs on the host emulator (CPU, small shapes) and, with -m gpu, on the MI355X through the same C ABI.
Tolerances: fp32 stream <= 2e-5 rel-L2 (north_star bar is 1e-3 on |STFT|); bf16 stream <= 1.5e-2 rel-L2."""
import pytest
import torch

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, NBSS_F32
from oracle import spatialnet_ref as ref
from util import Case, rel_l2

DTYPES = [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")]
# (B, F, T): small ragged shapes for the emulator; the gpu run adds the full BASELINE geometry
SHAPES = [(1, 5, 19), (2, 33, 40)]


MAX_SHAPES = [(1, 160, 3), (1, 2, 256)]  # the largest 8-kHz F (10 frequency tiles) and the longest training sequence (16 full strips)
BIG_F_SHAPES = [(1, 257, 3), (1, 272, 2)]  # 16 kHz (n_fft 512 -> 257 bins) and the largest F check_cfg accepts (17 frequency tiles): cross-band kernels


def shapes_for(backend):
    return SHAPES + MAX_SHAPES + ([(2, 129, 251)] if backend.name == "hip" else [])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kperm", [0, 1])
def test_mma_fragment_layout(backend, dtype, kperm):
    g = torch.Generator().manual_seed(3)
    A = torch.randn(16, 32, generator=g)
    Bm = torch.randn(32, 16, generator=g)
    if dtype == NBSS_BF16:
        A, Bm = A.bfloat16().float(), Bm.bfloat16().float()
    D = ops.selftest_mma(backend.lib, dtype, kperm, A.to(backend.device), Bm.to(backend.device))
    assert rel_l2(D, A.double() @ Bm.double()) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_encoder_decoder(backend, dtype):
    for (B, F, T) in shapes_for(backend):
        cs = Case(backend, B, F, T, dtype)
        xin, xin64 = cs.stream(seed=5, H=12)
        y = ops.encoder_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, xin)
        assert rel_l2(y, ref.encoder(xin64, cs.p64)) < cs.tol
        x, x64 = cs.stream(seed=6)
        o = ops.decoder_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, x)
        assert rel_l2(o, ref.decoder(x64, cs.p64)) < cs.tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("which", [0, 1])
def test_fconv(backend, dtype, which):
    for (B, F, T) in shapes_for(backend) + BIG_F_SHAPES + ([(2, 257, 126)] if backend.name == "hip" else []):
        cs = Case(backend, B, F, T, dtype)
        x, x64 = cs.stream(seed=7)
        y = ops.fconv_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, which, x)
        want = ref.fconv(x64, cs.p64, f"layers.0.fconv{which + 1}")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol  # the branch itself, not just x


@pytest.mark.parametrize("dtype", DTYPES)
def test_full(backend, dtype):
    for (B, F, T) in shapes_for(backend) + BIG_F_SHAPES + ([(2, 257, 126)] if backend.name == "hip" else []):
        cs = Case(backend, B, F, T, dtype)
        x, x64 = cs.stream(seed=8)
        y = ops.full_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
        want = ref.full(x64, cs.p64, "layers.0")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_mhsa(backend, dtype):
    # (1, 1, 251): the emulator also runs the full-length specialisation (all 16 key tiles, compile-time tile counts)
    for (B, F, T) in shapes_for(backend) + ([(1, 1, 251)] if backend.name != "hip" and dtype == NBSS_BF16 else []):
        cs = Case(backend, B, F, T, dtype)
        x, x64 = cs.stream(seed=9)
        y = ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
        want = ref.mhsa(x64, cs.p64, "layers.0")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol
        # the save buffer backward reads: attention output before out_proj, then fp32 log2-sum-exp rows [B,F,T,heads]
        save = ops.mhsa_save(cs.lib, cs.cfg, x.device)
        y2 = ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=save)
        assert torch.equal(y2, y)
        _, o_want, lse_want = ref.mhsa(x64, cs.p64, "layers.0", return_saved=True)
        n = B * F * T
        esz = x.element_size()
        o_got = save[: n * 96 * esz].view(x.dtype).view(B, F, T, 96)
        lse_off = (n * 96 * esz + 255) // 256 * 256
        lse_got = save[lse_off: lse_off + n * 4 * 4].view(torch.float32).view(B, F, T, 4)
        assert rel_l2(o_got, o_want) < cs.tol
        assert (lse_got.double().cpu() - lse_want).abs().max() < (2e-4 if dtype == NBSS_F32 else 5e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_tconvffn(backend, dtype):
    for (B, F, T) in shapes_for(backend) + ([(1, 1, 251)] if backend.name != "hip" and dtype == NBSS_BF16 else []):  # all 16 strips on the emulator too
        cs = Case(backend, B, F, T, dtype)
        x, x64 = cs.stream(seed=10)
        y = ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
        want = ref.tconvffn(x64, cs.p64, "layers.0")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol


# ---- sequences beyond 256 frames (forward only: validate / test / predict on full-length utterances) ------------------------
# emulator: one ragged chunk boundary each (tconvffn: 250-frame chunks; attention: 128-key blocks); gpu: T = 600 and 1001
def long_shapes(backend):
    return [(1, 1, 300), (1, 2, 257)] + ([(2, 129, 600), (1, 33, 1001)] if backend.name == "hip" else [])


@pytest.mark.parametrize("dtype", DTYPES)
def test_mhsa_long(backend, dtype):
    for (B, F, T) in long_shapes(backend):
        cs = Case(backend, B, F, T, dtype)
        x, x64 = cs.stream(seed=19)
        scratch = ops.mhsa_save(cs.lib, cs.cfg, x.device)  # beyond 256 frames this buffer is the K | V scratch
        y = ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=scratch)
        want = ref.mhsa(x64, cs.p64, "layers.0")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol
        with pytest.raises(RuntimeError):  # no scratch: refused, not silently wrong
            ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)


@pytest.mark.parametrize("dtype", DTYPES)
def test_tconvffn_long(backend, dtype):
    for (B, F, T) in long_shapes(backend):
        cs = Case(backend, B, F, T, dtype)
        x, x64 = cs.stream(seed=20)
        y = ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
        want = ref.tconvffn(x64, cs.p64, "layers.0")
        assert rel_l2(y, want) < cs.tol
        assert rel_l2(y.double().cpu() - x64, want - x64) < 3 * cs.tol


def test_long_sequences_are_forward_only(backend):
    cs = Case(backend, 1, 2, 300, NBSS_BF16)
    x, _ = cs.stream(seed=21)
    grads = torch.zeros_like(cs.flat)
    with pytest.raises(RuntimeError):
        ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, grads, cs.packed, 0, x, x, ops.workspace(cs.lib, cs.cfg, x.device))


@pytest.mark.parametrize("dtype", DTYPES)
def test_network_long(backend, dtype):
    """whole network, inference, T beyond 256 through nbss_spatialnet_fwd (ping-pong buffers, attention scratch at the head of ws)"""
    from nbss_amd.engine import SpatialNetEngine
    from nbss_amd._lib import NbssError
    B, F, T, L = (1, 129, 600, 8) if backend.name == "hip" else (1, 2, 270, 2)
    p = ref.init_params(num_layers=L, num_freqs=F, dim_input=12, dim_output=4, seed=4)
    eng = SpatialNetEngine(backend.lib, backend.device, dim_input=12, dim_output=4, num_freqs=F, num_layers=L, dtype=dtype)
    eng.load_params(p)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, F, T, 12, generator=g).to(eng.stream_dtype())
    y = eng.forward(x.to(backend.device), train=False)
    want = ref.spatialnet(x.double(), {k: v.double() for k, v in p.items()}, L)
    assert rel_l2(y, want) < (1e-4 if dtype == NBSS_F32 else 3e-2)
    with pytest.raises(NbssError):
        eng.forward(x.to(backend.device), train=True)
