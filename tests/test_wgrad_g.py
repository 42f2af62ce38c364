"""The tile weight gradient of the dense maps of the geometry-generic path (nbss_amd/csrc/wgrad_g.hip; NBSS_WGRAD_TILE=0 selects the other path, read once per process)
against the column-slice path of wgrad.hip it replaces: the attention block's and the T-ConvFFN block's parameter gradients of SpatialNet-large from two
child processes, a token count that is not a multiple of the 32-token chunk."""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import sys, torch
sys.path.insert(0, {tests!r}); sys.path.insert(0, {root!r})
from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, Lib, hip
from nbss_amd.build import build_emu
from conftest import Backend
from util import Case
be = Backend("hip", hip(), torch.device("cuda:0")) if {gpu} else Backend("emu", Lib(build_emu()), torch.device("cpu"))
cs = Case(be, {B}, {F}, {T}, NBSS_BF16, geo="large")
x, _ = cs.stream(seed=1); dy, _ = cs.stream(seed=2, scale=0.5)
ws = ops.workspace(cs.lib, cs.cfg, be.device)
G = torch.zeros_like(cs.flat)
o = ops.mhsa_save(cs.lib, cs.cfg, be.device)
ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=o)
ops.mhsa_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, o, ws)
ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws)
torch.save(G.cpu(), {out!r})
"""


def _run(tmp_path, gpu, B, F, T):
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / f"g{flag}.pt")
        code = CHILD.format(tests=str(ROOT / "tests"), root=str(ROOT), gpu=gpu, B=B, F=F, T=T, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NBSS_WGRAD_TILE=flag), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(out))
    tile, ref = outs
    assert int((ref != 0).sum()) > 400000  # in_proj, out_proj and both FFN maps (+ biases, convs, norms) were written
    assert float((tile - ref).norm() / ref.norm()) < 2e-6


@pytest.mark.parametrize("shape", [(1, 3, 45), (1, 1, 5), (1, 2, 16), (2, 1, 33)], ids=lambda v: "x".join(map(str, v)))
def test_tile_weight_gradient_equals_the_slices_emu(tmp_path, shape):
    """135 tokens = 4 chunks + 7 rows; fewer tokens than one chunk; exactly one chunk; 66 = 2 chunks + 2 rows"""
    _run(tmp_path, False, *shape)


@pytest.mark.gpu
def test_tile_weight_gradient_equals_the_slices_hip(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    _run(tmp_path, True, 2, 129, 251)  # 64 758 tokens: 2 023 chunks + 22 rows over 85 / 128 / 256 token shares
