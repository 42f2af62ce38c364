"""Golden vectors produced BY THE REFERENCE (tests/golden/make_golden.py, models.arch.SpatialNet / models.io.*):
  * the oracle restatement must reproduce them (pins the oracle), and
  * the HIP kernels must reproduce them through the C ABI (emulator on CPU, libnbss_hip.so with -m gpu).
fp32 reference outputs: <= 1e-4 rel-L2 for the fp32 stream (north-star bar 1e-3), <= 2e-2 for the bf16 stream."""
from pathlib import Path

import numpy as np
import pytest
import torch

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, NBSS_F32
from nbss_amd.engine import SpatialNetEngine
from oracle import io_ref
from oracle import spatialnet_ref as ref
from util import rel_l2

G = Path(__file__).resolve().parent / "golden"


def load_net():
    z = np.load(G / "spatialnet_F9_T21_L2.npz")
    p = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    g = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad/")}
    return z, p, g


def test_oracle_reproduces_reference_network():
    z, p, g = load_net()
    x, L = torch.from_numpy(z["x"]).double(), int(z["L"])
    leaves, p64 = {}, {}
    for k, v in p.items():
        key = "layers.0.full." + k.rsplit(".", 1)[1] if ".full." in k else k  # `full` is one shared tensor
        if key not in leaves:
            leaves[key] = p[key].double().clone().requires_grad_(True)
        p64[k] = leaves[key]
    y = ref.spatialnet(x, p64, L)
    assert rel_l2(y, torch.from_numpy(z["y"])) < 2e-6
    (y * torch.from_numpy(z["r"]).double()).sum().backward()
    for k, want in g.items():
        assert rel_l2(leaves[k].grad, want) < 2e-5, k


def test_oracle_reproduces_reference_stft_norm():
    z = np.load(G / "stft_norm_n1500.npz")
    sig = torch.from_numpy(z["sig"]).double()
    X = io_ref.stft(sig)
    assert rel_l2(X.real, z["X_re"]) < 1e-6 and rel_l2(X.imag, z["X_im"]) < 1e-6
    Xn, mm = io_ref.norm_frequency_online(X, 1)
    assert rel_l2(mm, z["XrMM"]) < 1e-6
    assert rel_l2(io_ref.istft(X, 1500), z["back"]) < 1e-6


@pytest.mark.parametrize("dtype", [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")])
def test_kernels_reproduce_reference_network(backend, dtype):
    z, p, g = load_net()
    F, T, L = int(z["F"]), int(z["T"]), int(z["L"])
    eng = SpatialNetEngine(backend.lib, backend.device, dim_input=12, dim_output=4, num_freqs=F, num_layers=L, dtype=dtype)
    eng.load_params(p)
    x = torch.from_numpy(z["x"]).to(eng.stream_dtype()).to(backend.device)
    y = eng.forward(x, train=True)
    tol = 1e-4 if dtype == NBSS_F32 else 2e-2
    assert rel_l2(y, z["y"]) < tol
    eng.backward(x, torch.from_numpy(z["r"]).to(backend.device))
    views = eng.param_views(eng.grads)
    gtol = 5e-4 if dtype == NBSS_F32 else 8e-2
    bad = [(k, rel_l2(views[k], want)) for k, want in g.items() if rel_l2(views[k], want) > gtol]
    assert not bad, bad


def test_kernels_reproduce_reference_stft_norm(backend):
    z = np.load(G / "stft_norm_n1500.npz")
    sig = torch.from_numpy(z["sig"]).to(backend.device)
    tab = ops.stft_tables(backend.lib, 256, 0, backend.device)
    X, mm = ops.stft_norm_fwd(backend.lib, 256, NBSS_F32, tab, sig, 1)
    assert rel_l2(mm, z["XrMM"][:, 0]) < 2e-5
    assert rel_l2(X.cpu().double() * torch.from_numpy(z["XrMM"][:, 0]).double()[..., None], z["Xl"].astype(np.float64) * z["XrMM"][:, 0][..., None]) < 2e-5
    assert rel_l2(X, z["Xl"]) < 1e-3
