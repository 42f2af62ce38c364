"""Pin the oracle against the reference's OWN modules, imported read-only from /root/reference (build container only;
skipped on the GPU box where the reference tree does not exist)."""
import importlib
import sys

import pytest
import torch

from oracle import io_ref
from oracle import spatialnet_ref as ref
from util import rel_l2

REF = "/root/reference"


@pytest.fixture()
def reference_modules(reference_available):
    if not reference_available:
        pytest.skip("reference tree not present")
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REF)
    try:
        mods = {n: importlib.import_module(n) for n in ("models.arch.SpatialNet", "models.io.stft", "models.io.norm")}
        assert REF in mods["models.arch.SpatialNet"].__file__
        yield mods
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_spatialnet_small_forward_and_grads(reference_modules):
    """SpatialNet-small exactly as configs/SpatialNet.yaml builds it (1 191 092 parameters), short input"""
    torch.manual_seed(0)
    torch.set_num_threads(4)
    SpatialNet = reference_modules["models.arch.SpatialNet"].SpatialNet
    net = SpatialNet(dim_input=12, dim_output=4, num_layers=8, dim_hidden=96, dim_ffn=192, num_heads=4, kernel_size=(5, 3), conv_groups=(8, 8),
                     norms=("LN", "LN", "GN", "LN", "LN", "LN"), dim_squeeze=8, num_freqs=129, full_share=0).double()
    assert sum(p.numel() for p in net.parameters()) == 1191092  # images/model_size_and_flops.png: 1.2 M
    x = torch.randn(1, 129, 30, 12, dtype=torch.float64)
    y = net(x)
    p = {k: v.detach() for k, v in net.state_dict().items()}
    assert rel_l2(ref.spatialnet(x, p, 8), y) < 1e-10
    # each sub-block against the reference layer's own methods
    lay = net.layers[3]
    h = torch.randn(1, 129, 30, 96, dtype=torch.float64)
    assert rel_l2(ref.fconv(h, p, "layers.3.fconv1"), h + lay._fconv(lay.fconv1, h)) < 1e-10
    assert rel_l2(ref.full(h, p, "layers.3"), h + lay._full(h)) < 1e-10
    lay.need_weights = False
    assert rel_l2(ref.mhsa(h, p, "layers.3"), h + lay._tsa(h, None)[0]) < 1e-10
    assert rel_l2(ref.tconvffn(h, p, "layers.3"), h + lay._tconvffn(h)) < 1e-10


def test_spatialnet_large_16khz_forward(reference_modules):
    """SpatialNet-large as the "for large" comments of configs/SpatialNet.yaml build it (12 layers, 192 / 384 / squeeze 16), 257 frequencies
    (16 kHz): the oracle the large-geometry HIP kernels are tested against (tests/test_large.py) IS the reference's forward"""
    torch.manual_seed(1)
    torch.set_num_threads(4)
    SpatialNet = reference_modules["models.arch.SpatialNet"].SpatialNet
    net = SpatialNet(dim_input=12, dim_output=4, num_layers=12, dim_hidden=192, dim_ffn=384, num_heads=4, kernel_size=(5, 3), conv_groups=(8, 8),
                     norms=("LN", "LN", "GN", "LN", "LN", "LN"), dim_squeeze=16, num_freqs=257, full_share=0).double()
    x = torch.randn(1, 257, 12, 12, dtype=torch.float64)
    p = {k: v.detach() for k, v in net.state_dict().items()}
    assert rel_l2(ref.spatialnet(x, p, 12), net(x)) < 1e-10


def test_state_dict_interchanges_with_the_reference(reference_modules, tmp_path):
    """checkpoint contract: the drop-in module's state_dict loads strictly into the reference's SpatialNet and
    back, through a file written by SharedTrainer.save_checkpoint"""
    import importlib.util
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    RefNet = reference_modules["models.arch.SpatialNet"].SpatialNet
    kw = dict(dim_input=12, dim_output=4, num_layers=3, dim_hidden=96, dim_ffn=192, num_heads=4, kernel_size=(5, 3), conv_groups=(8, 8),
              norms=("LN", "LN", "GN", "LN", "LN", "LN"), dim_squeeze=8, num_freqs=129, full_share=0)
    spec = importlib.util.spec_from_file_location("dropin_spatialnet", root / "models" / "arch" / "SpatialNet.py")
    mod = importlib.util.module_from_spec(spec)
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    for k in saved:  # the drop-in's own relative imports must resolve against THIS repo, not the reference on sys.path
        del sys.modules[k]
    sys.path.insert(0, str(root))
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(str(root))
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    torch.manual_seed(3)
    ours = mod.SpatialNet(**kw)
    refnet = RefNet(**kw)
    sd = ours.state_dict()
    assert list(sd.keys()) == list(refnet.state_dict().keys())
    refnet.load_state_dict(sd, strict=True)           # ours -> reference
    torch.manual_seed(4)
    ours2 = mod.SpatialNet(**kw)
    ours2.load_state_dict(refnet.state_dict(), strict=True)  # reference -> ours
    for (k, a), (_, b) in zip(sd.items(), ours2.state_dict().items()):
        assert torch.equal(a, b), k
    # and the oracle evaluated on these weights is the reference's forward (so the HIP path, pinned to the oracle, is too)
    x = torch.randn(1, 129, 12, 12, dtype=torch.float64)
    p = {k: v.detach().double() for k, v in sd.items()}
    assert rel_l2(ref.spatialnet(x, p, 3), refnet.double()(x)) < 1e-10


def test_stft_norm_istft(reference_modules):
    STFT, Norm = reference_modules["models.io.stft"].STFT, reference_modules["models.io.norm"].Norm
    stft, norm = STFT(n_fft=256, n_hop=128), Norm(mode="frequency")
    x = torch.randn(2, 4, 3000)
    X, n = stft.stft(x)
    assert rel_l2(io_ref.stft(x), X) < 1e-6
    Xn, (Xr, XrMM) = norm.norm(X.clone(), ref_channel=2)
    on, omm = io_ref.norm_frequency_online(X, 2)
    assert rel_l2(on, Xn) < 1e-6 and rel_l2(omm, XrMM) < 1e-6
    assert rel_l2(io_ref.istft(X, n), stft.istft(X, n)) < 1e-6
    assert rel_l2(stft.istft(X, n), x) < 1e-5  # the reference's own round-trip smoke check (stft.py:106-112)
