"""Bitwise repeatability of the training step (round 5): every parameter-gradient reduction is a fixed-order fold — per-workgroup partial tiles /
rows summed by ONE owner per element (wgrad_reduce_kernel, tailw_finalize_kernel + tailw_affine_kernel, affine_slices/final_kernel, tconv_part_reduce),
per-wave slots instead of LDS float atomics inside fconv_bwd / full_bwd — so two runs from the same state produce the same bits, in order and with the
walks' second stream.  (Rounds 1-4: parameter gradients repeatable to ~3e-7, the last fold of each reduction used float atomics.)"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from nbss_amd._lib import NBSS_BF16  # noqa: E402
from nbss_amd.engine import SpatialNetEngine, TrainStep  # noqa: E402
from oracle import spatialnet_ref as ref  # noqa: E402


def _run(lib, dev, steps, B, L, N):
    eng = SpatialNetEngine(lib, dev, dim_input=12, dim_output=4, num_freqs=129, num_layers=L, dtype=NBSS_BF16)
    eng.load_params(ref.init_params(num_layers=L, num_freqs=129, dim_input=12, dim_output=4, seed=0))
    ts = TrainStep(eng, lr=1e-3, clip=5.0)
    g = torch.Generator().manual_seed(5)
    grads, losses = [], []
    for i in range(steps):
        x = torch.randn(B, 6, N, generator=g).to(dev)
        yr = torch.randn(B, 2, N, generator=g).to(dev)
        if i == 0:  # the flat gradient of the first step, before the optimizer consumes it
            loss, _, dout, xin, _ = ts.forward_loss(x, yr, need_grad=True)
            eng.backward(xin, dout)
            torch.cuda.synchronize()
            grads.append(eng.grads.clone())
            ts.apply_gradients(reduced=True)
            losses.append(float(loss))
        else:
            losses.append(float(ts.step(x, yr)))
    torch.cuda.synchronize()
    return grads[0].cpu().numpy(), eng.params.cpu().numpy(), np.array(losses)


def _check(lib):
    dev = torch.device("cuda:0")
    g0, p0, l0 = _run(lib, dev, 4, 4, 2, 32000)
    g1, p1, l1 = _run(lib, dev, 4, 4, 2, 32000)
    assert np.isfinite(l0).all() and np.abs(g0).max() > 0
    bad = np.flatnonzero(g0.view(np.uint32) != g1.view(np.uint32))
    assert bad.size == 0, (bad.size, bad[:8], g0[bad[:8]], g1[bad[:8]])
    assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
    assert np.array_equal(l0, l1)


@pytest.mark.gpu
def test_training_step_is_bitwise_repeatable(hip_lib):
    _check(hip_lib)  # the default: the walks' gradient launches on the library's second stream


@pytest.mark.gpu
def test_training_step_is_bitwise_repeatable_in_order():
    """the same with NBSS_SIDE_STREAM=0 (read once per process: a child process)"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    env = dict(os.environ, NBSS_SIDE_STREAM="0")
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, cwd=str(Path(__file__).resolve().parent.parent), timeout=600)
    assert r.returncode == 0 and "repeatable" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


if __name__ == "__main__":
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from nbss_amd._lib import hip
    _check(hip())
    print("repeatable")
