"""The dense tile GEMM of the geometry-generic path (nbss_amd/csrc/gemm_g.hip: 128 x 192 workgroup tiles through an LDS-DMA ring) against torch, through the
one-operation C-ABI calls that dispatch to it (bf16, one tap, one group, K % 32 == 0): ragged row counts, output widths that are not a multiple of the
192-wide chunk, bias / SiLU / residual / second-output epilogues."""
import pytest
import torch
import torch.nn.functional as Fn

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16
from util import rel_l2


def _lin(backend, nseq, T, K, M, res, act_out, seed, y2=False):
    lib, dev = backend.lib, backend.device
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(nseq, T, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(M, K, 1, generator=g) / K ** 0.5).to(dev)
    b = (torch.randn(M, generator=g) * 0.1).to(dev)
    r = torch.randn(nseq, T, M, generator=g).to(torch.bfloat16).to(dev) if res else None
    y = torch.full((nseq, T, M), float("nan"), dtype=torch.bfloat16, device=dev)
    want = Fn.linear(x.double().cpu(), w[..., 0].double().cpu(), b.double().cpu())
    if y2:
        h = torch.full_like(y, float("nan"))
        ws = torch.empty(lib._dll.nbss_nb_bwd_ws_bytes(M, K, 1, 1), dtype=torch.uint8, device=dev)
        lib.call("nbss_nb_conv_t_train", NBSS_BF16, nseq, T, K, K, M, 1, 1, ops._ptr(lib, x), ops._ptr(lib, w), ops._ptr(lib, b), ops._ptr(lib, y), ops._ptr(lib, h),
                 None, ops._ptr(lib, ws), ops._stream(lib, x))
        assert rel_l2(y, want) < 1.5e-2
        assert rel_l2(h, Fn.silu(y.double().cpu())) < 1.5e-2
        return
    ws = torch.empty(lib._dll.nbss_nb_ws_bytes(M, K, 1, 1), dtype=torch.uint8, device=dev)
    lib.call("nbss_nb_conv_t", NBSS_BF16, nseq, T, K, K, M, 1, 1, ops._ptr(lib, x), ops._ptr(lib, w), ops._ptr(lib, b), ops._ptr(lib, y),
             ops._ptr(lib, r) if res else None, 0, act_out, ops._ptr(lib, ws), ops._stream(lib, x))
    if act_out:
        want = Fn.silu(want)
    if res:
        want = want + r.double().cpu()
    assert torch.isfinite(y.float()).all()
    assert rel_l2(y, want) < 1.5e-2


CASES = [  # nseq, T, K, M, residual, act_out
    (3, 100, 192, 576, False, 0),   # in_proj of SpatialNet-large: 3 output chunks, 300 rows = 2.3 row tiles
    (1, 129, 384, 192, True, 0),    # FFN map back with a residual
    (2, 67, 192, 384, False, 1),    # SiLU on the output
    (1, 50, 64, 80, False, 0),      # fewer outputs than a chunk, two K-slabs
    (5, 251, 576, 192, False, 0),   # in_proj data gradient: 18 K-slabs, 1 255 rows
    (1, 7, 96, 208, True, 1),       # a second chunk with one valid output tile, 7 rows
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_dense_tile_gemm(backend, case):
    nseq, T, K, M, res, act = case
    _lin(backend, nseq, T, K, M, res, act, seed=K + M)


def test_dense_tile_gemm_second_output(backend):
    _lin(backend, 2, 90, 192, 384, False, 0, seed=5, y2=True)


def test_dense_tile_gemm_many_tiles_per_workgroup(backend):
    """more tiles than the 512 persistent workgroups: every workgroup walks several tiles, the ring runs across tile boundaries"""
    if backend.name == "emu":
        pytest.skip("hip only: 1 100 row tiles x 3 chunks")
    _lin(backend, 561, 251, 192, 576, False, 0, seed=9)
