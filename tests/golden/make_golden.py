"""Generate the golden fixtures FROM THE REFERENCE ITSELF (run in the build container, where /root/reference exists):

    python tests/golden/make_golden.py

Imports the reference's own modules (models.arch.SpatialNet.SpatialNet, models.io.stft.STFT, models.io.norm.Norm) on CPU,
fp32, and stores small seeded input/parameter/output triples as .npz next to this script.  The fixtures travel to the GPU
box (where /root/reference does not exist) and pin both the oracle and the HIP kernels.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def main():
    assert REF.exists(), "the reference tree is needed to (re)generate the fixtures"
    # the repo root shadows `models`: import the reference's package explicitly from its own root
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[m]
    sys.path.insert(0, str(REF))
    from models.arch.SpatialNet import SpatialNet  # noqa: E402  (reference)
    from models.io.norm import Norm  # noqa: E402
    from models.io.stft import STFT  # noqa: E402
    assert "/root/reference" in sys.modules["models.arch.SpatialNet"].__file__

    torch.manual_seed(0)
    torch.set_num_threads(1)
    F, T, L = 9, 21, 2
    net = SpatialNet(dim_input=12, dim_output=4, num_layers=L, dim_hidden=96, dim_ffn=192, kernel_size=(5, 3), conv_groups=(8, 8),
                     norms=("LN", "LN", "GN", "LN", "LN", "LN"), dim_squeeze=8, num_freqs=F, num_heads=4, full_share=0).eval()
    with torch.no_grad():  # make every affine / bias term non-trivial
        for n, p in net.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    x = torch.randn(1, F, T, 12)
    x.requires_grad_(False)
    y = net(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    out = {"x": x.numpy(), "y": y.detach().numpy(), "r": r.numpy(), "F": F, "T": T, "L": L}
    sd = net.state_dict()
    for k, v in sd.items():
        out["param/" + k] = v.numpy()
    for n, p in net.named_parameters():  # shared `full` parameters appear once here (remove_duplicate)
        out["grad/" + n] = p.grad.numpy()
    np.savez_compressed(HERE / "spatialnet_F9_T21_L2.npz", **out)

    # STFT + Norm('frequency') + layout glue and the iSTFT round trip (models/io/stft.py, norm.py; SharedTrainer.py:113-131)
    stft, norm = STFT(n_fft=256, n_hop=128), Norm(mode="frequency")
    sig = torch.randn(2, 3, 1500)
    X, n = stft.stft(sig)
    Xn, (Xr, XrMM) = norm.norm(X.clone(), ref_channel=1)
    Xl = torch.view_as_real(Xn.permute(0, 2, 3, 1)).reshape(2, 129, X.shape[-1], 6)
    back = stft.istft(X, n)
    np.savez_compressed(HERE / "stft_norm_n1500.npz", sig=sig.numpy(), X_re=X.real.numpy(), X_im=X.imag.numpy(), Xl=Xl.numpy(),
                        XrMM=XrMM.numpy(), back=back.numpy())
    print("written:", [p.name for p in HERE.glob("*.npz")])


if __name__ == "__main__":
    main()
