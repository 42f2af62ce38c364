"""The cross-band `full` kernels pick their slab width (8 / 4 / 2 frames per workgroup) from the grid that results (full.hip: full_tt): small
test shapes only ever see the narrowest.  NBSS_FULL_TT forces a width (read once per process), so the forward / backward parity tests of the
block are re-run in a child process per width; on the GPU the same is done against the real library."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(width, marker):
    env = dict(os.environ, NBSS_FULL_TT=str(width))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_kernels_fwd.py", "tests/test_kernels_bwd.py", "-q", "-x", "-m", marker, "-k", "full",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("width", [4, 8])
def test_full_block_parity_at_every_slab_width_emu(width):
    _run(width, "not gpu")


@pytest.mark.gpu
@pytest.mark.parametrize("width", [2, 4, 8])
def test_full_block_parity_at_every_slab_width_hip(width):
    _run(width, "gpu")
