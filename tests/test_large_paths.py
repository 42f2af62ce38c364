"""The kernels SpatialNet-large's bf16 backward runs on by default (tchain.hip: conv chain, fconv_g.hip: F-conv block) against the one-pass-per-operation
path of gbwd.hip they replaced (NBSS_TCHAIN_OFF=1 / NBSS_FCONVG_OFF=1, read once per process): the same block from two child processes on the emulator.
Both paths are tested against the oracle in test_large.py; this pins them to each other (the unfused path stays the fallback for T > 256 / F > 160 / fp32)."""
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
from nbss_amd._lib import NBSS_BF16, Lib
from nbss_amd.build import build_emu
from conftest import Backend
from util import Case
be = Backend("emu", Lib(build_emu()), torch.device("cpu"))
cs = Case(be, {B}, {F}, {T}, NBSS_BF16, geo="large")
x, _ = cs.stream(seed=1); dy, _ = cs.stream(seed=2, scale=0.5)
ws = ops.workspace(cs.lib, cs.cfg, be.device)
G = torch.zeros_like(cs.flat)
if {block!r} == "tconvffn":
    dx = ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws)
else:
    dx = ops.fconv_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, 1, x, dy, ws)
torch.save((dx.float().cpu(), G.cpu()), {out!r})
"""


def _pair(tmp_path, block, env, B, F, T):
    res = []
    for flag in ("0", "1"):
        out = str(tmp_path / f"{block}{flag}.pt")
        code = CHILD.format(tests=str(ROOT / "tests"), root=str(ROOT), B=B, F=F, T=T, block=block, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{env: flag}), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(torch.load(out))
    return res


def test_conv_chain_kernel_vs_unfused_path(tmp_path):
    (dx, G), (dx0, G0) = _pair(tmp_path, "tconvffn", "NBSS_TCHAIN_OFF", 2, 3, 40)
    assert not torch.equal(dx, dx0)  # (two different kernels ran)
    # intermediate tensors are rounded to bf16 at different points (dh5 is stored before SiLU' is applied, dh4 is not stored at all)
    assert float((dx - dx0).norm() / dx0.norm()) < 6e-3
    assert float((G - G0).norm() / G0.norm()) < 8e-3


def test_fconv_slab_kernel_vs_unfused_path(tmp_path):
    (dx, G), (dx0, G0) = _pair(tmp_path, "fconv", "NBSS_FCONVG_OFF", 2, 37, 5)
    # the same roundings in the same places: equal up to the order of the fp32 sums
    assert float((dx - dx0).norm() / dx0.norm()) < 1e-4
    assert float((G - G0).norm() / G0.norm()) < 1e-5


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 2, 3), (1, 1, 17), (1, 2, 256), (2, 1, 255)], ids=lambda v: "x".join(map(str, v)))
def test_conv_chain_kernel_edge_shapes(tmp_path, shape):
    """single frames, sequences shorter than a conv's reach, one frame past a 16-row tile, the longest sequence the kernel takes and one less"""
    (dx, G), (dx0, G0) = _pair(tmp_path, "tconvffn", "NBSS_TCHAIN_OFF", *shape)
    assert torch.isfinite(dx).all() and torch.isfinite(G).all()
    assert float((dx - dx0).norm() / dx0.norm()) < 8e-3
    assert float((G - G0).norm() / G0.norm()) < 1.2e-2


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 2, 2), (2, 5, 1), (1, 17, 3), (1, 144, 2)], ids=lambda v: "x".join(map(str, v)))
def test_fconv_slab_kernel_edge_shapes(tmp_path, shape):
    """one frequency (every tap but the centre is padding), fewer frequencies than taps, one past a tile, nine full tiles"""
    (dx, G), (dx0, G0) = _pair(tmp_path, "fconv", "NBSS_FCONVG_OFF", *shape)
    assert torch.isfinite(dx).all() and torch.isfinite(G).all()
    assert float((dx - dx0).norm() / dx0.norm()) < 1e-4
    assert float((G - G0).norm() / max(float(G0.norm()), 1e-30)) < 1e-5
