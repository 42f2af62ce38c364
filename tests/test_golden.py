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


def load_large():
    z = np.load(G / "spatialnet_large_F17_T21_L1.npz")
    p = {k[6:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("param/")}  # (2-D+ weights stored in fp16)
    return z, p


def test_oracle_reproduces_reference_large_network():
    """SpatialNet-large built BY THE REFERENCE (192 / 384 / squeeze 16, 4 heads of 48): pins the oracle's geometry-generic restatement"""
    z, p = load_large()
    y = ref.spatialnet(torch.from_numpy(z["x"]).double(), {k: v.double() for k, v in p.items()}, int(z["L"]))
    assert rel_l2(y, torch.from_numpy(z["y"])) < 2e-6


@pytest.mark.parametrize("dtype", [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")])
def test_kernels_reproduce_reference_large_network(backend, dtype):
    """the large-geometry forward kernels against the reference's own output (fp32 stream <= 1e-4, north-star bar 1e-3)"""
    z, p = load_large()
    F, L = int(z["F"]), int(z["L"])
    eng = SpatialNetEngine(backend.lib, backend.device, dim_input=12, dim_output=4, num_freqs=F, num_layers=L, dtype=dtype, dim_hidden=192, dim_ffn=384,
                           dim_squeeze=16)
    eng.load_params(p)
    x = torch.from_numpy(z["x"]).to(eng.stream_dtype()).to(backend.device)
    y = eng.forward(x, train=False)
    assert rel_l2(y, z["y"]) < (1e-4 if dtype == NBSS_F32 else 2e-2)


def test_oracle_and_kernels_reproduce_reference_stft_16khz(backend):
    """n_fft 512 / hop 256 (paras_16k of the reference's models/io/stft.py): STFT + per-bin norm + layout glue, iSTFT round trip"""
    z = np.load(G / "stft_norm_n3000_16k.npz")
    sig64 = torch.from_numpy(z["sig"]).double()
    X = io_ref.stft(sig64, 512, 256)
    assert rel_l2(X.real, z["X_re"]) < 1e-6 and rel_l2(X.imag, z["X_im"]) < 1e-6
    assert rel_l2(io_ref.istft(X, 3000, 512, 256), z["back"]) < 1e-6
    sig = torch.from_numpy(z["sig"]).to(backend.device)
    tab = ops.stft_tables(backend.lib, 512, 0, backend.device)
    Xk, mm = ops.stft_norm_fwd(backend.lib, 512, NBSS_F32, tab, sig, 1)
    assert rel_l2(mm, z["XrMM"][:, 0]) < 2e-5
    assert rel_l2(Xk, z["Xl"]) < 1e-3
    # inverse: the reference's iSTFT of its own (un-normalised) spectrum, through inorm + iSTFT kernels fed with X / XrMM
    mm_t = torch.from_numpy(z["XrMM"][:, 0]).to(backend.device)
    Xc = torch.complex(torch.from_numpy(z["X_re"]), torch.from_numpy(z["X_im"]))  # [B, C, F, T]
    out = torch.view_as_real((Xc[:, :2] / torch.from_numpy(z["XrMM"])).permute(0, 2, 3, 1).contiguous()).reshape(2, 257, Xc.shape[-1], 4).float()
    y = ops.inorm_istft_fwd(backend.lib, 512, tab, out.to(backend.device).contiguous(), mm_t.contiguous(), 3000)
    assert rel_l2(y, z["back"][:, :2]) < 2e-5
