"""bf16 parity pinned against what the REFERENCE itself loses in bf16.

tests/golden/bf16_reference_errors.json (tests/golden/make_golden.py bf16ref) holds, per tensor, the rel-L2 error of the reference's own
SpatialNet under torch.autocast(bfloat16) — Lightning's bf16-mixed, the precision the headline metric is quoted at — against the same
module in fp64.  The bf16 stream of this repo, on the same inputs and parameters, against the fp64 oracle, must be no worse than 1.5x
that per tensor (with a small floor for tensors where both are at rounding level) and inside this repo's absolute bars.
Stated bf16 tolerances: output <= 1.5e-2 rel-L2 (the reference's own: 6.1e-3 .. 8.5e-3); parameter gradients: every tensor <= 1.5 x the
reference's own bf16 deviation on that tensor, the worst tensor and the median <= 1.1 x the reference's own worst (0.115) / median
(2.9e-2 at 2 layers, 5.7e-2 at 8)."""
import json
import statistics
from pathlib import Path

import numpy as np
import pytest
import torch

from nbss_amd._lib import NBSS_BF16
from nbss_amd.engine import SpatialNetEngine
from oracle import spatialnet_ref as ref
from util import rel_l2

G = Path(__file__).resolve().parent / "golden"
REFERR = json.loads((G / "bf16_reference_errors.json").read_text())
FLOOR = 5e-3  # both sides at bf16 rounding level: ratios are noise


def _oracle(p, x, r, L):
    leaves, p64 = {}, {}
    for k, v in p.items():
        if id(v) not in leaves:
            leaves[id(v)] = v.double().clone().requires_grad_(True)
        p64[k] = leaves[id(v)]
    y = ref.spatialnet(x.double(), p64, L)
    (y * r.double()).sum().backward()
    return y.detach(), {k: p64[k].grad for k in p64}


def _check(case, backend, p, x, r, F, L):
    eng = SpatialNetEngine(backend.lib, backend.device, dim_input=12, dim_output=4, num_freqs=F, num_layers=L, dtype=NBSS_BF16)
    eng.load_params(p)
    xs = x.to(torch.bfloat16).to(backend.device)
    y = eng.forward(xs, train=True)
    eng.backward(xs, r.float().to(backend.device))
    views = eng.param_views(eng.grads)
    wy, wg = _oracle(p, xs.float().cpu(), r, L)  # (the oracle sees the bf16-rounded input, as the stream does)
    want = REFERR[case]
    ey = rel_l2(y, wy)
    errs = {k: rel_l2(views[k], g) for k, g in wg.items() if k in want["grads"]}
    worse = {k: (round(e, 4), round(want["grads"][k], 4)) for k, e in errs.items() if e > max(1.5 * want["grads"][k], FLOOR)}
    print(f"{case}: y {ey:.3e} (reference bf16: {want['y']:.3e}); grads median {statistics.median(errs.values()):.3e} (reference "
          f"{statistics.median(want['grads'].values()):.3e}), max {max(errs.values()):.3e} (reference {max(want['grads'].values()):.3e})")
    assert ey <= max(1.5 * want["y"], FLOOR) and ey <= 1.5e-2, (ey, want["y"])
    assert not worse, worse
    # worst tensor / median pinned to the REFERENCE's own worst / median on this case (x 1.1: two bf16 computations of the same network differ by
    # that much in their noisiest tensor), not to an absolute figure tuned to the current kernels
    assert max(errs.values()) <= 1.1 * max(want["grads"].values()), (max(errs.values()), max(want["grads"].values()))
    assert statistics.median(errs.values()) <= 1.1 * statistics.median(want["grads"].values())


def test_bf16_stream_is_within_the_references_own_bf16_error_small(backend):
    """F 9 / T 21 / 2 layers: the reference-generated fixture's x, r and parameters"""
    z = np.load(G / "spatialnet_F9_T21_L2.npz")
    p = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    full = {}
    for k in list(p):  # `full` is one shared tensor under every layer's key
        if ".full." in k:
            p[k] = full.setdefault(k.rsplit(".", 1)[1], p["layers.0.full." + k.rsplit(".", 1)[1]])
    _check("F9_T21_L2", backend, p, torch.from_numpy(z["x"]), torch.from_numpy(z["r"]), int(z["F"]), int(z["L"]))


@pytest.mark.gpu
def test_bf16_stream_is_within_the_references_own_bf16_error_headline_width(hip_lib):
    """129 frequencies, 64 frames, 8 layers: parameters / inputs regenerated from the seeds the fixture was made with (checksum-pinned)"""
    from conftest import Backend
    c = REFERR["F129_T64_L8"]
    p = ref.init_params(num_layers=8, num_freqs=129, seed=c["seeds"]["init_params"])
    g = torch.Generator().manual_seed(c["seeds"]["x_r"])
    x = torch.randn(1, 129, 64, 12, generator=g)
    r = torch.randn(1, 129, 64, 4, generator=g)
    flat = torch.cat([v.double().reshape(-1) for v in p.values()])
    got = [float(flat.sum()), float(flat.abs().sum()), float(x.double().sum()), float(r.double().sum())]
    assert np.allclose(got, c["checksum"], rtol=1e-9), "the seeded parameters differ from the ones the reference errors were measured on"
    _check("F129_T64_L8", Backend("hip", hip_lib, torch.device("cuda:0")), p, x, r, 129, 8)
