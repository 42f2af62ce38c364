"""The walks' second stream (csrc/side.h).  Backward: every parameter-gradient launch runs on a library-owned gradient stream behind a fork
event, with one workspace copy per sub-block kind, three rotating gradient buffers and `done` events guarding what gets overwritten.  Forward:
the last, mostly empty round of sequences of the attention -> T-ConvFFN pair is a second launch on that stream.  A missing dependency shows up
as a gradient that differs from the in-order walk (NBSS_SIDE_STREAM=0, read once per process: child processes), or that changes from step to
step on identical inputs."""
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from nbss_amd._lib import NBSS_BF16, hip
from nbss_amd.engine import SpatialNetEngine
from oracle import spatialnet_ref as ref
B, L, reps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev = torch.device("cuda:0")
eng = SpatialNetEngine(hip(), dev, dim_input=12, dim_output=4, num_freqs=129, num_layers=L, dtype=NBSS_BF16)
eng.load_params(ref.init_params(num_layers=L, num_freqs=129, dim_input=12, dim_output=4, seed=0))
g = torch.Generator().manual_seed(3)
xin = torch.randn(B, 129, 251, 12, generator=g).to(dev).to(torch.bfloat16)
dout = torch.randn(B, 129, 251, 4, generator=g).to(dev)
outs, grads = [], []
for r in range(reps):
    eng.grads.zero_()
    y = eng.forward(xin, train=True)
    eng.backward(xin, dout)
    torch.cuda.synchronize()
    outs.append(y.float().cpu().numpy()); grads.append(eng.grads.cpu().numpy().copy())
np.savez(out, y=np.stack(outs), g=np.stack(grads))
"""


def _child(env_side, B, L, reps, path):
    env = dict(os.environ)
    if env_side is None:
        env.pop("NBSS_SIDE_STREAM", None)
    else:
        env["NBSS_SIDE_STREAM"] = env_side
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=str(ROOT)), str(B), str(L), str(reps), path], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return np.load(path)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [2, 8])  # 258 and 1032 sequences: a tail launch of 2 resp. 8 sequences; slab widths 2 and 8 of the full block
def test_two_stream_walk_equals_in_order_walk(B):
    with tempfile.TemporaryDirectory() as d:
        two = _child(None, B, 3, 4, os.path.join(d, "two.npz"))
        one = _child("0", B, 3, 1, os.path.join(d, "one.npz"))
    # the forward output is bit-identical (same kernels per sequence, no reductions across launches)
    assert np.array_equal(two["y"][0], one["y"][0])
    for r in range(1, two["y"].shape[0]):
        assert np.array_equal(two["y"][r], two["y"][0])
    # parameter gradients: the last folds use float atomics (order-dependent rounding, <= 3e-7 relative per DESIGN §5)
    ref = one["g"][0].astype(np.float64)
    scale = np.linalg.norm(ref)
    assert scale > 0 and np.isfinite(ref).all()
    for r in range(two["g"].shape[0]):
        diff = np.linalg.norm(two["g"][r].astype(np.float64) - ref) / scale
        assert diff <= 2e-5, (r, diff)
        worst = np.max(np.abs(two["g"][r].astype(np.float64) - ref)) / (np.max(np.abs(ref)) + 1e-30)
        assert worst <= 1e-4, (r, worst)
