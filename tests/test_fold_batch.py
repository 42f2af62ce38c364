"""Folds of the partial parameter gradients, batched (csrc/fold.h: one table-kernel launch per dependency stage of a sub-block, small grids) vs one launch
per fold (NBSS_FOLD_BATCH=0, what large grids run): the bodies are the same device functions (csrc/foldk.h) and every sum keeps its order, so the two must
agree BIT FOR BIT — data gradients and every parameter gradient of every backward sub-block, both stream types.  NBSS_FOLD_BATCH is read once per process:
two worker processes (tests/emu_schedule_worker.py: every sub-block once, outputs saved)."""
import os
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent


def _run(flag: str, tmp: Path, pool: str = "") -> dict:
    out = tmp / f"fold_{flag}_{pool}.pt"
    env = dict(os.environ, NBSS_FOLD_BATCH=flag, HIPEMU_ORDER="fwd")
    if pool:
        env["NBSS_FOLD_POOL"] = pool
    subprocess.run([sys.executable, str(ROOT / "tests" / "emu_schedule_worker.py"), str(out)], check=True, env=env, cwd=str(ROOT), timeout=1500)
    return torch.load(out)


def test_batched_folds_equal_single_launches_bitwise(tmp_path):
    from nbss_amd.build import build_emu
    build_emu()  # once, before the workers race to build it
    from concurrent.futures import ThreadPoolExecutor
    # pools of 400 KB / 100 KB: the scopes run full in mid sub-block (flush, then reuse of the pool) resp. meet requests larger than the pool (those
    # folds leave alone, behind everything pending) — the partial tiles of the worker's shapes are 40 - 320 KB per launch
    with ThreadPoolExecutor(4) as ex:
        single, batched, tight, tiny = ex.map(lambda a: _run(a[0], tmp_path, a[1]), (("0", ""), ("1", ""), ("1", "400000"), ("1", "100000")))
    grads = [k for k in single if k.endswith("_G")]
    assert len(grads) >= 10 and all(float(single[k].abs().max()) > 0 for k in grads)
    for name, got in (("batched", batched), ("pool 400 KB", tight), ("pool 100 KB", tiny)):
        bad = [(k, int((single[k] != got[k]).sum())) for k in single if not torch.equal(single[k], got[k])]
        assert not bad, (name, bad)


def _device_worker(out: str):
    """child process on the GPU box: two training steps of a 2-layer network at batch 2, the flat gradient / parameters / losses saved"""
    import numpy as np
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    from nbss_amd._lib import hip
    from test_determinism import _run as run_steps
    g, p, l = run_steps(hip(), torch.device("cuda:0"), 2, 2, 2, 32000)
    np.savez(out, g=g, p=p, l=l)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_batched_folds_equal_single_launches_on_the_device(tmp_path):
    import numpy as np
    res = []
    for flag in ("0", "1"):
        out = tmp_path / f"dev_{flag}.npz"
        env = dict(os.environ, NBSS_FOLD_BATCH=flag)
        r = subprocess.run([sys.executable, __file__, str(out)], env=env, capture_output=True, text=True, cwd=str(ROOT), timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res.append(np.load(out))
    a, b = res
    assert np.abs(a["g"]).max() > 0 and np.isfinite(a["l"]).all()
    for k in ("g", "p", "l"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


if __name__ == "__main__":
    _device_worker(sys.argv[1])
