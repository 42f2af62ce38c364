"""Backward parity of every SpatialNet sub-block: dx and ALL parameter gradients of the HIP kernels
against torch.autograd through the fp64 oracle, for the same seeded x and upstream gradient dy.
fp32 stream: <= 1e-4 rel-L2 per tensor; bf16 stream: <= 3e-2 rel-L2 per tensor (gradients of sums
over thousands of bf16-rounded terms)."""
import pytest
import torch

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, NBSS_F32
from nbss_amd.params import param_table
from oracle import spatialnet_ref as ref
from util import Case, rel_l2

DTYPES = [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")]
SHAPES = [(1, 1, 1), (1, 5, 19), (2, 33, 40)]


MAX_SHAPES = [(1, 160, 3), (1, 2, 256), (1, 130, 4)]  # the largest F and T check_cfg accepts (10 frequency tiles / 16 full strips); 9 frequency tiles (the 9-wave fconv_bwd)


BIG_F_SHAPES = [(1, 257, 3), (1, 272, 2)]  # 16 kHz (n_fft 512 -> 257 bins), 17 frequency tiles: the cross-band kernels, bf16 stream (fp32 backward stops at F = 160)


def big_f_shapes(backend, dtype):
    return (BIG_F_SHAPES + ([(2, 257, 126)] if backend.name == "hip" else [])) if dtype == NBSS_BF16 else []


def shapes_for(backend):
    return SHAPES + MAX_SHAPES + ([(2, 129, 251)] if backend.name == "hip" else [])


def oracle_grads(fn, x64, p64, dy64, names):
    """autograd of sum(fn(x, p) * dy) w.r.t. x and the named parameters (fp64)"""
    x = x64.clone().requires_grad_(True)
    p = dict(p64)
    leaves = {}
    for n in names:
        if id(p64[n]) in leaves:
            p[n] = leaves[id(p64[n])]
        else:
            p[n] = p64[n].clone().requires_grad_(True)
            leaves[id(p64[n])] = p[n]
    y = fn(x, p)
    (y * dy64).sum().backward()
    return x.grad, {n: p[n].grad for n in names}


def check_param_grads(cs, G, want, tol):
    table = param_table(cs.lib, cs.cfg)
    bad = []
    for n, g in want.items():
        off, shape = table[n]
        got = G[off:off + g.numel()].reshape(shape)
        scale = float(g.abs().max())
        err = rel_l2(got, g) if scale > 0 else float(got.abs().max())
        if err > tol:
            bad.append((n, err))
    assert not bad, bad


def run_block_bwd(backend, dtype, B, F, T, fwd_ref, bwd_op, names, seed, bf16_tol=3e-2):
    cs = Case(backend, B, F, T, dtype)
    x, x64 = cs.stream(seed=seed)
    dy, dy64 = cs.stream(seed=seed + 100, scale=0.5)
    G = torch.zeros_like(cs.flat)
    ws = ops.workspace(cs.lib, cs.cfg, backend.device)
    dx = bwd_op(cs, G, x, dy, ws)
    want_dx, want_g = oracle_grads(lambda xx, pp: fwd_ref(xx, pp), x64, cs.p64, dy64, names)
    tol = 1e-4 if dtype == NBSS_F32 else bf16_tol
    assert rel_l2(dx, want_dx) < tol, ("dx", rel_l2(dx, want_dx))
    assert rel_l2(dx.double().cpu() - dy64, want_dx - dy64) < 3 * tol  # the branch gradient itself
    check_param_grads(cs, G, want_g, tol)


TF_NAMES = [f"layers.0.tconvffn.{i}.{wb}" for i in (0, 1, 3, 5, 6, 8, 10) for wb in ("weight", "bias")]


@pytest.mark.parametrize("dtype", DTYPES)
def test_tconvffn_bwd(backend, dtype):
    for (B, F, T) in shapes_for(backend) + ([(1, 1, 251)] if backend.name != "hip" and dtype == NBSS_BF16 else []):  # all 16 strips on the emulator too
        run_block_bwd(backend, dtype, B, F, T, lambda x, p: ref.tconvffn(x, p, "layers.0"),
                      lambda cs, G, x, dy, ws: ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws), TF_NAMES, seed=20)


def test_tconvffn_bwd_from_saved_preactivations(backend):
    """bf16 stream: the training-mode forward keeps its four pre-activations + LayerNorm / GroupNorm statistics; the backward kernel that
    reads them (no forward recompute, the three T-conv weight gradients contracted in-kernel) meets the same bars as the recomputing one,
    and the forward output is bitwise the inference forward's"""
    shapes = [(1, 5, 19), (2, 33, 40), (1, 2, 256), (1, 3, 70)] + ([(2, 129, 251)] if backend.name == "hip" else [(1, 1, 251)])
    for (B, F, T) in shapes:
        saved = {}

        def bwd(cs, G, x, dy, ws):
            sv = ops.tconvffn_save(cs.lib, cs.cfg, backend.device)
            assert sv is not None
            y = ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, t_save=sv)
            saved["same"] = torch.equal(y, ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x))
            return ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws, t_save=sv)

        run_block_bwd(backend, NBSS_BF16, B, F, T, lambda x, p: ref.tconvffn(x, p, "layers.0"), bwd, TF_NAMES, seed=20)
        assert saved["same"]


def test_tconvffn_save_is_refused_where_backward_recomputes(backend):
    cs = Case(backend, 1, 5, 19, NBSS_F32)
    assert ops.tconvffn_save(cs.lib, cs.cfg, backend.device) is None  # fp32 stream: nbss_tconvffn_save_bytes == 0


MH_NAMES = ["layers.0.norm_mhsa.weight", "layers.0.norm_mhsa.bias", "layers.0.mhsa.in_proj_weight", "layers.0.mhsa.in_proj_bias",
            "layers.0.mhsa.out_proj.weight", "layers.0.mhsa.out_proj.bias"]


@pytest.mark.parametrize("dtype", DTYPES)
def test_mhsa_bwd(backend, dtype):
    # (1, 1, 251): the emulator also runs the full-length specialisation (all 16 strips, compile-time tile counts)
    for (B, F, T) in shapes_for(backend) + ([(1, 1, 251)] if backend.name != "hip" and dtype == NBSS_BF16 else []):
        def bwd(cs, G, x, dy, ws):
            o = ops.mhsa_save(cs.lib, cs.cfg, x.device)
            ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=o)
            return ops.mhsa_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, o, ws)

        run_block_bwd(backend, dtype, B, F, T, lambda x, p: ref.mhsa(x, p, "layers.0"), bwd, MH_NAMES, seed=30)


def test_mhsa_bwd_sequence_lengths(emu_lib):
    """the single-sweep attention backward (bf16) across the sequence-length cases of its loops: one frame, partial / odd numbers of 16-frame strips,
    odd numbers of query pairs and of 64-query dS chunks, the boundaries of the full-length specialisation (240 | 241) and its last frame counts"""
    from conftest import Backend
    be = Backend("emu", emu_lib, torch.device("cpu"))

    def bwd(cs, G, x, dy, ws):
        o = ops.mhsa_save(cs.lib, cs.cfg, x.device)
        ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=o)
        return ops.mhsa_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, o, ws)

    for T in (1, 17, 33, 65, 129, 240, 241, 256):
        run_block_bwd(be, NBSS_BF16, 1, 2 if T < 64 else 1, T, lambda x, p: ref.mhsa(x, p, "layers.0"), bwd, MH_NAMES, seed=30 + T)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("which", [0, 1])
def test_fconv_bwd(backend, dtype, which):
    pre = f"layers.0.fconv{which + 1}"
    names = [f"{pre}.0.weight", f"{pre}.0.bias", f"{pre}.1.weight", f"{pre}.1.bias", f"{pre}.2.weight"]
    for (B, F, T) in shapes_for(backend) + big_f_shapes(backend, dtype):
        run_block_bwd(backend, dtype, B, F, T, lambda x, p: ref.fconv(x, p, pre),
                      lambda cs, G, x, dy, ws: ops.fconv_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, which, x, dy, ws), names, seed=40 + which,
                      bf16_tol=5e-2)  # PReLU kink: bf16 rounding flips the sign of ~1% of the pre-activations


FULL_NAMES = ["layers.0.norm_full.weight", "layers.0.norm_full.bias", "layers.0.squeeze.0.weight", "layers.0.squeeze.0.bias",
              "layers.0.full.weight", "layers.0.full.bias", "layers.0.unsqueeze.0.weight", "layers.0.unsqueeze.0.bias"]


@pytest.mark.parametrize("dtype", DTYPES)
def test_full_bwd(backend, dtype):
    # (fp32 stream at F > 160: served by narrower slabs — full.hip: full_bwd_width — unlike the F-conv backward, whose images do not fit at any width)
    for (B, F, T) in shapes_for(backend) + (big_f_shapes(backend, dtype) or BIG_F_SHAPES):
        run_block_bwd(backend, dtype, B, F, T, lambda x, p: ref.full(x, p, "layers.0"),
                      lambda cs, G, x, dy, ws: ops.full_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws), FULL_NAMES, seed=50)


@pytest.mark.parametrize("dtype", DTYPES)
def test_encoder_decoder_bwd(backend, dtype):
    for (B, F, T) in shapes_for(backend):
        cs = Case(backend, B, F, T, dtype)
        tol = 1e-4 if dtype == NBSS_F32 else 3e-2
        ws = ops.workspace(cs.lib, cs.cfg, backend.device)
        # decoder
        x, x64 = cs.stream(seed=60)
        g = torch.Generator().manual_seed(61)
        dout = torch.randn(B, F, T, 4, generator=g)
        G = torch.zeros_like(cs.flat)
        dx = ops.decoder_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, x, dout.to(backend.device), ws)
        want_dx, want_g = oracle_grads(lambda xx, pp: ref.decoder(xx, pp), x64, cs.p64, dout.double(), ["decoder.weight", "decoder.bias"])
        assert rel_l2(dx, want_dx) < tol
        check_param_grads(cs, G, want_g, tol)
        # encoder (parameter gradients only)
        xin, xin64 = cs.stream(seed=62, H=12)
        dy, dy64 = cs.stream(seed=63)
        G = torch.zeros_like(cs.flat)
        ops.encoder_bwd(cs.lib, cs.cfg, G, xin, dy)
        _, want_g = oracle_grads(lambda xx, pp: ref.encoder(xx, pp), xin64, cs.p64, dy64, ["encoder.weight", "encoder.bias"])
        check_param_grads(cs, G, want_g, tol)


def test_fp32_backward_stops_at_160_frequencies(backend):
    """the fp32-stream images of the F-conv backward kernel do not fit the LDS beyond F = 160: refused loudly, not computed wrongly
    (the full-band block narrows its slabs instead: test_full_bwd)"""
    from nbss_amd._lib import NbssError
    cs = Case(backend, 1, 257, 2, NBSS_F32)
    x, _ = cs.stream(seed=1)
    dy, _ = cs.stream(seed=2)
    G = torch.zeros_like(cs.flat)
    ws = ops.workspace(cs.lib, cs.cfg, backend.device)
    with pytest.raises(NbssError, match="UNSUPPORTED"):
        ops.fconv_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, 0, x, dy, ws)


def test_block_forward_backward_random_small_grids(emu_lib):
    """every fused block, forward and backward, both stream types, on random tiny grids (emulator): single frames / frequencies and sizes around the
    16-row tiles, the conv kernels' reach and the slab widths — the shapes the fixed cases above do not visit"""
    from hypothesis import given, settings, strategies as st
    from conftest import Backend
    be = Backend("emu", emu_lib, torch.device("cpu"))
    mh = ["layers.0.norm_mhsa.weight", "layers.0.norm_mhsa.bias", "layers.0.mhsa.in_proj_weight", "layers.0.mhsa.in_proj_bias", "layers.0.mhsa.out_proj.weight",
          "layers.0.mhsa.out_proj.bias"]
    fc = [f"layers.0.fconv1.{k}" for k in ("0.weight", "0.bias", "1.weight", "1.bias", "2.weight")]
    blocks = {"fconv": (lambda x, p: ref.fconv(x, p, "layers.0.fconv1"), lambda cs, x: ops.fconv_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, 0, x),
                        lambda cs, G, x, dy, ws: ops.fconv_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, 0, x, dy, ws), fc, 5e-2),
              "full": (lambda x, p: ref.full(x, p, "layers.0"), lambda cs, x: ops.full_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x),
                       lambda cs, G, x, dy, ws: ops.full_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws), FULL_NAMES, 3e-2),
              "mhsa": (lambda x, p: ref.mhsa(x, p, "layers.0"), None, None, mh, 3e-2),
              "tconvffn": (lambda x, p: ref.tconvffn(x, p, "layers.0"), lambda cs, x: ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x),
                           lambda cs, G, x, dy, ws: ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws), TF_NAMES, 3e-2)}

    @settings(max_examples=16, deadline=None)
    @given(B=st.integers(1, 2), F=st.sampled_from([1, 2, 3, 5, 17]), T=st.sampled_from([1, 2, 3, 5, 16, 17, 33]), block=st.sampled_from(sorted(blocks)),
           dtype=st.sampled_from([NBSS_F32, NBSS_BF16]))
    def check(B, F, T, block, dtype):
        run(B, F, T, block, dtype)

    def run(B, F, T, block, dtype):
        fwd_ref, fwd_op, bwd_op, names, btol = blocks[block]
        cs = Case(be, B, F, T, dtype)
        x, x64 = cs.stream(seed=11)
        dy, dy64 = cs.stream(seed=111, scale=0.5)
        G = torch.zeros_like(cs.flat)
        ws = ops.workspace(cs.lib, cs.cfg, be.device)
        if block == "mhsa":
            save = ops.mhsa_save(cs.lib, cs.cfg, be.device)
            y = ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=save)
            dx = ops.mhsa_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, save, ws)
        else:
            y = fwd_op(cs, x)
            dx = bwd_op(cs, G, x, dy, ws)
        tol = 1e-4 if dtype == NBSS_F32 else btol
        assert rel_l2(y, fwd_ref(x64, cs.p64)) < (2e-5 if dtype == NBSS_F32 else 1.5e-2), ("fwd", B, F, T, block, dtype)
        want_dx, want_g = oracle_grads(fwd_ref, x64, cs.p64, dy64, names)
        assert rel_l2(dx, want_dx) < tol, ("dx", B, F, T, block, dtype, rel_l2(dx, want_dx))
        # (a handful of tokens: one bf16 pre-activation on the other side of the PReLU kink is a visible share of a parameter's gradient)
        check_param_grads(cs, G, want_g, tol if T * F * B >= 128 or dtype == NBSS_F32 else 4 * tol)  # (sweep_small_grids: (2,3,16) fconv bf16 lands at 5.1e-2)

    # every block on the single-token grid and on the other one-workgroup grids (round-5 review: full_bwd's fold scratch ran into the weight-gradient
    # partial tiles at B F T = 1 and four parameter gradients came back zero), then the seeded draw
    for block in sorted(blocks):
        for dtype in (NBSS_F32, NBSS_BF16):
            for (B, F, T) in [(1, 1, 1), (1, 1, 2), (2, 1, 1), (1, 2, 1)]:
                run(B, F, T, block, dtype)
    check()
