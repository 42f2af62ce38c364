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


def large_16k():
    """SpatialNet-large (the "for large" comments of configs/SpatialNet.yaml: 192 / 384 / squeeze 16) and the 16-kHz STFT setting
    (n_fft 512, hop 256): `python tests/golden/make_golden.py large16k` writes only these two fixtures."""
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[m]
    sys.path.insert(0, str(REF))
    from models.arch.SpatialNet import SpatialNet  # noqa: E402  (reference)
    from models.io.norm import Norm  # noqa: E402
    from models.io.stft import STFT  # noqa: E402
    assert "/root/reference" in sys.modules["models.arch.SpatialNet"].__file__
    torch.manual_seed(7)
    torch.set_num_threads(1)
    F, T, L = 17, 21, 1
    net = SpatialNet(dim_input=12, dim_output=4, num_layers=L, dim_hidden=192, dim_ffn=384, kernel_size=(5, 3), conv_groups=(8, 8),
                     norms=("LN", "LN", "GN", "LN", "LN", "LN"), dim_squeeze=16, num_freqs=F, num_heads=4, full_share=0).eval()
    with torch.no_grad():
        for n, p in net.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        x = torch.randn(1, F, T, 12)
        y = net(x)
    out = {"x": x.numpy(), "y": y.numpy(), "F": F, "T": T, "L": L}
    for k, v in net.state_dict().items():
        out["param/" + k] = v.numpy().astype(np.float16) if v.dim() > 1 else v.numpy()  # (weights stored in fp16: the loader widens them, the fixture's y was computed from the widened values below)
    # recompute y from the fp16-rounded weights so that the fixture is self-consistent
    with torch.no_grad():
        sd = {k: (v.half().float() if v.dim() > 1 else v) for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        out["y"] = net(x).numpy()
    np.savez_compressed(HERE / "spatialnet_large_F17_T21_L1.npz", **out)
    stft, norm = STFT(n_fft=512, n_hop=256), Norm(mode="frequency")
    sig = torch.randn(2, 3, 3000)
    X, n = stft.stft(sig)
    Xn, (Xr, XrMM) = norm.norm(X.clone(), ref_channel=1)
    Xl = torch.view_as_real(Xn.permute(0, 2, 3, 1)).reshape(2, 257, X.shape[-1], 6)
    back = stft.istft(X, n)
    np.savez_compressed(HERE / "stft_norm_n3000_16k.npz", sig=sig.numpy(), X_re=X.real.numpy(), X_im=X.imag.numpy(), Xl=Xl.numpy(),
                        XrMM=XrMM.numpy(), back=back.numpy())
    print("written: spatialnet_large_F17_T21_L1.npz, stft_norm_n3000_16k.npz")


def bf16_reference_errors():
    """What the REFERENCE's own bf16-mixed computation loses against fp64 (`python tests/golden/make_golden.py bf16ref`):
    the reference's SpatialNet under torch.autocast('cpu', torch.bfloat16) — Lightning's `bf16-mixed`: fp32 parameters, bf16 convolutions /
    linears / attention — forward + backward of sum(y * r), per-tensor rel-L2 error of the output and of every parameter gradient against
    the same module in fp64.  Two cases: the F9 / T21 / L2 fixture (its stored x, r, parameters) and a 129-frequency, 64-frame, 8-layer
    network whose parameters the oracle's seeded init_params generates (so that the test can regenerate them on the GPU box; a checksum
    pins them).  Only the error figures are stored (bf16_reference_errors.json): the test asserts that the HIP bf16 stream is no worse
    than 1.5x the reference's own bf16 deviation, tensor by tensor.  (CPU autocast; the CUDA autocast policy additionally keeps
    layer_norm in fp32, so the reference's GPU figure can only be smaller in the norm-heavy tensors — the bar stays conservative.)"""
    import json
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[m]
    sys.path.insert(0, str(REF))
    from models.arch.SpatialNet import SpatialNet  # noqa: E402  (reference)
    assert "/root/reference" in sys.modules["models.arch.SpatialNet"].__file__
    sys.path.insert(0, str(HERE.parent.parent))
    from oracle import spatialnet_ref as oref  # (parameter generator only)

    def rel(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / b.norm()) if float(b.norm()) > 0 else float(a.norm())

    def run(F, T, L, sd, x, r, **geo):
        kw = dict(dim_input=12, dim_output=4, num_layers=L, dim_hidden=96, dim_ffn=192, kernel_size=(5, 3), conv_groups=(8, 8),
                  norms=("LN", "LN", "GN", "LN", "LN", "LN"), dim_squeeze=8, num_freqs=F, num_heads=4, full_share=0)
        kw.update(geo)
        res = {}
        for mode in ("fp64", "bf16"):
            net = SpatialNet(**kw).eval()
            net.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
            if mode == "fp64":
                net = net.double()
                y = net(x.double())
                (y * r.double()).sum().backward()
            else:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    y = net(x.float())
                (y.float() * r.float()).sum().backward()
            res[mode] = (y.detach(), {n: p.grad.detach() for n, p in net.named_parameters()})
        y64, g64 = res["fp64"]
        ybf, gbf = res["bf16"]
        return {"y": rel(ybf, y64), "grads": {n: rel(gbf[n], g64[n]) for n in g64}}

    out = {"how": "tests/golden/make_golden.py bf16ref; rel-L2 of the reference SpatialNet under torch.autocast('cpu', bfloat16) vs the same module in fp64",
           "torch": torch.__version__}
    torch.set_num_threads(8)
    z = np.load(HERE / "spatialnet_F9_T21_L2.npz")
    sd = {k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    out["F9_T21_L2"] = run(9, 21, 2, sd, torch.from_numpy(z["x"]), torch.from_numpy(z["r"]))
    p = oref.init_params(num_layers=8, num_freqs=129, seed=5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 129, 64, 12, generator=g)
    r = torch.randn(1, 129, 64, 4, generator=g)
    c = run(129, 64, 8, p, x, r)
    flat = torch.cat([v.double().reshape(-1) for v in p.values()])
    c["seeds"] = {"init_params": 5, "x_r": 6}
    c["checksum"] = [float(flat.sum()), float(flat.abs().sum()), float(x.double().sum()), float(r.double().sum())]
    out["F129_T64_L8"] = c
    # SpatialNet-large (configs/SpatialNet.yaml "for large" comments: 192 / 384 / squeeze 16, 12 layers) at the headline grid 129 x 251:
    # what tests/test_large.py's bf16 bars for the large train step are set from (`bf16ref large` regenerates only this case)
    if "large" in sys.argv[2:] or "all" in sys.argv[2:]:
        geo = dict(dim_hidden=192, dim_ffn=384, dim_squeeze=16)
        pl = oref.init_params(num_layers=12, num_freqs=129, seed=7, **geo)
        g = torch.Generator().manual_seed(8)
        xl = torch.randn(1, 129, 251, 12, generator=g)
        rl = torch.randn(1, 129, 251, 4, generator=g)
        cl = run(129, 251, 12, pl, xl, rl, **geo)
        flat = torch.cat([v.double().reshape(-1) for v in pl.values()])
        cl["seeds"] = {"init_params": 7, "x_r": 8}
        cl["checksum"] = [float(flat.sum()), float(flat.abs().sum()), float(xl.double().sum()), float(rl.double().sum())]
        out["large_F129_T251_L12"] = cl
        print("large: y", cl["y"], "worst grads", sorted(cl["grads"].items(), key=lambda kv: -kv[1])[:6])
    else:
        prev = json.loads((HERE / "bf16_reference_errors.json").read_text())
        if "large_F129_T251_L12" in prev:
            out["large_F129_T251_L12"] = prev["large_F129_T251_L12"]
    (HERE / "bf16_reference_errors.json").write_text(json.dumps(out, indent=1))
    worst = sorted(c["grads"].items(), key=lambda kv: -kv[1])[:6]
    print("written: bf16_reference_errors.json; L8 case: y", c["y"], "worst grads", worst)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "bf16ref":
        assert REF.exists(), "the reference tree is needed to (re)generate the fixtures"
        return bf16_reference_errors()
    if len(sys.argv) > 1 and sys.argv[1] == "online96":
        assert REF.exists(), "the reference tree is needed to (re)generate the fixtures"
        return online_native_width()
    if len(sys.argv) > 1 and sys.argv[1] == "nbnative":
        assert REF.exists(), "the reference tree is needed to (re)generate the fixtures"
        return nb_models_native_width()
    if len(sys.argv) > 1 and sys.argv[1] == "large16k":
        assert REF.exists(), "the reference tree is needed to (re)generate the fixtures"
        return large_16k()
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
    nb_models()
    online_models()
    print("written:", [p.name for p in HERE.glob("*.npz")])


def online_models():
    """tiny seeded OnlineSpatialNet instances of the reference (OnlineSpatialNet.py, base/retention.py): banded causal MHSA and
    retention (parallel form), eval mode: input, state_dict, output, gradients of sum(y * r).
    The attention window (31) covers the whole fixture sequence (21 frames): with the torch of this image the reference's call
    `mhsa(x, x, x, attn_mask=band, is_causal=True, need_weights=False)` takes nn.MultiheadAttention's causal fast path, which DROPS the
    band mask — the reference attends to all past frames whatever N is; the two only agree (and the fixture is only meaningful) for T <= N."""
    from models.arch.OnlineSpatialNet import OnlineSpatialNet  # noqa: E402  (reference)
    assert "/root/reference" in sys.modules["models.arch.OnlineSpatialNet"].__file__
    # mamba_ssm is not installed: the reference then sets Mamba = None and its `isinstance(self.mhsa, Mamba)` raises for EVERY attention
    # type; a placeholder class lets the mhsa / retention variants run unmodified
    sys.modules["models.arch.OnlineSpatialNet"].Mamba = type("Mamba", (), {})
    torch.manual_seed(3)
    out = {}
    for name, kw in (("mhsa31", dict(attention="mhsa(31)")), ("ret2", dict(attention="ret(2)", decay=[4, 5, 9, 10], rope=False))):
        net = OnlineSpatialNet(dim_input=4, dim_output=4, num_layers=2, dim_squeeze=8, num_freqs=9, encoder_kernel_size=5, dim_hidden=32, dim_ffn=64,
                               num_heads=4, dropout=(0, 0, 0), kernel_size=(5, 3), conv_groups=(8, 8), norms=["LN", "LN", "GN", "LN", "LN", "LN"],
                               full_share=0, **kw).eval()
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
        x = torch.randn(2, 9, 21, 4)
        y = net(x)
        r = torch.randn_like(y)
        (y * r).sum().backward()
        out[f"{name}/x"], out[f"{name}/y"], out[f"{name}/r"] = x.numpy(), y.detach().numpy(), r.numpy()
        for k, v in net.state_dict().items():
            out[f"{name}/param/{k}"] = v.numpy()
        for k, p in net.named_parameters():
            out[f"{name}/grad/{k}"] = p.grad.numpy()
    np.savez_compressed(HERE / "online_tiny.npz", **out)
    online_native_width()


def online_native_width():
    """`python tests/golden/make_golden.py online96`: the reference's OnlineSpatialNet at the width the NATIVE streaming step serves (dim_hidden 96,
    dim_ffn 192, 4 heads; 9 frequencies, 24 frames, 2 layers, batch 2), ret(2) and a banded mhsa whose window covers the sequence: input, state_dict
    and the whole-utterance output — tests/test_online.py runs the HIP step chunk by chunk against THIS output (one hop from the reference)."""
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[m]
    if str(REF) not in sys.path or sys.path[0] != str(REF):
        sys.path.insert(0, str(REF))
    from models.arch.OnlineSpatialNet import OnlineSpatialNet  # noqa: E402  (reference)
    assert "/root/reference" in sys.modules["models.arch.OnlineSpatialNet"].__file__
    sys.modules["models.arch.OnlineSpatialNet"].Mamba = type("Mamba", (), {})
    torch.manual_seed(13)
    out = {}
    for name, kw in (("ret2", dict(attention="ret(2)", decay=[4, 5, 9, 10], rope=False)), ("mhsa31", dict(attention="mhsa(31)"))):
        net = OnlineSpatialNet(dim_input=4, dim_output=4, num_layers=2, dim_squeeze=8, num_freqs=9, encoder_kernel_size=5, dim_hidden=96, dim_ffn=192,
                               num_heads=4, dropout=(0, 0, 0), kernel_size=(5, 3), conv_groups=(8, 8), norms=["LN", "LN", "GN", "LN", "LN", "LN"],
                               full_share=0, **kw).eval()
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
            x = torch.randn(2, 9, 24, 4)
            y = net(x)
        out[f"{name}/x"], out[f"{name}/y"] = x.numpy(), y.numpy()
        for k, v in net.state_dict().items():
            out[f"{name}/param/{k}"] = v.numpy().astype(np.float32)
    np.savez_compressed(HERE / "online_w96.npz", **out)
    print("written: online_w96.npz")


def nb_models():
    """tiny seeded instances of the reference's narrow-band models (blstm2_fc1.py, NBC2.py, NBC.py) and of the NBSS time-domain
    wrapper: input, every state_dict tensor, output and the gradient of sum(y * r) w.r.t. every parameter.  NBSS.py imports
    si_sdr / pit from torchmetrics (absent here) at module level only for its loss helper: a stub module lets it import; the
    forward pass stored below does not touch it."""
    import types
    tm = types.ModuleType("torchmetrics"); tmf = types.ModuleType("torchmetrics.functional"); tma = types.ModuleType("torchmetrics.functional.audio")
    tma.permutation_invariant_training = tma.scale_invariant_signal_distortion_ratio = lambda *a, **k: None
    sys.modules.update({"torchmetrics": tm, "torchmetrics.functional": tmf, "torchmetrics.functional.audio": tma})
    from models.arch.blstm2_fc1 import BLSTM2_FC1  # noqa: E402  (reference)
    from models.arch.NBC import NBC  # noqa: E402
    from models.arch.NBC2 import NBC2  # noqa: E402
    from models.arch.NBSS import NBSS  # noqa: E402
    assert "/root/reference" in sys.modules["models.arch.NBSS"].__file__
    torch.manual_seed(1)
    bk = {"n_heads": 2, "dropout": 0, "conv_kernel_size": 3, "n_conv_groups": 4, "norms": ("LN", "GBN", "GBN"),
          "group_batch_norm_kwargs": {"share_along_sequence_dim": False}}
    cases = {
        "blstm": (BLSTM2_FC1(dim_input=4, dim_output=4, hidden_size=(8, 6)), torch.randn(2, 5, 11, 4)),
        "nbc2": (NBC2(dim_input=4, dim_output=4, n_layers=2, dim_hidden=16, dim_ffn=32, num_freqs=5, block_kwargs=bk), torch.randn(2, 5, 11, 4)),
        "nbc": (NBC(dim_input=4, dim_output=4, n_layers=2, encoder_kernel_size=4, n_heads=4, hidden_size=16, ffn_size=32), torch.randn(2, 5, 11, 4)),
        "nbss": (NBSS(n_channel=2, n_speaker=2, n_fft=64, n_overlap=32, ref_channel=1, arch="NB_BLSTM", arch_kwargs={"hidden_size": (8, 6)}),
                 torch.randn(2, 2, 700)),
    }
    out = {}
    for name, (net, x) in cases.items():
        net.eval()  # NBC's blocks carry dropout 0.1 by default (NBC.py:170): eval mode makes the fixture deterministic
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
        y = net(x)
        r = torch.randn_like(y)
        (y * r).sum().backward()
        out[f"{name}/x"], out[f"{name}/y"], out[f"{name}/r"] = x.numpy(), y.detach().numpy(), r.numpy()
        for k, v in net.state_dict().items():
            out[f"{name}/param/{k}"] = v.numpy()
        for k, p in net.named_parameters():
            out[f"{name}/grad/{k}"] = p.grad.numpy()
    np.savez_compressed(HERE / "nb_models_tiny.npz", **out)


def nb_models_native_width():
    """The reference's three narrow-band networks at the SMALLEST widths the native HIP paths take (attention heads 24 wide, conv groups 8 / 16 wide, LSTM
    hidden size 128), run in fp64: input, parameters, output and the gradient of sum(y * r) w.r.t. every parameter -> nb_models_native.npz.  The native
    paths (nbss_amd/nbc2.py, nbc.py, blstm.py) are compared with THESE numbers directly (tests/test_nb_native_vs_reference.py), not through the repo's own
    torch.nn modules.  Not stored: NBC's sinusoid tables (rel_pos.pe: a constructor constant); of the BiLSTM's four big matrices every eighth row of the
    gradient (the parameters are rounded to fp16 values first and stored as fp16: 1 MB instead of 4)."""
    import types
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[m]
    sys.path.insert(0, str(REF))
    tm = types.ModuleType("torchmetrics"); tmf = types.ModuleType("torchmetrics.functional"); tma = types.ModuleType("torchmetrics.functional.audio")
    tma.permutation_invariant_training = tma.scale_invariant_signal_distortion_ratio = lambda *a, **k: None
    sys.modules.update({"torchmetrics": tm, "torchmetrics.functional": tmf, "torchmetrics.functional.audio": tma})
    from models.arch.blstm2_fc1 import BLSTM2_FC1  # noqa: E402  (reference)
    from models.arch.NBC import NBC  # noqa: E402
    from models.arch.NBC2 import NBC2  # noqa: E402
    assert "/root/reference" in sys.modules["models.arch.NBC2"].__file__
    torch.manual_seed(11)
    torch.set_num_threads(1)
    bk = {"n_heads": 2, "dropout": 0, "conv_kernel_size": 3, "n_conv_groups": 4, "norms": ("LN", "GBN", "GBN"),
          "group_batch_norm_kwargs": {"share_along_sequence_dim": False}}
    cases = {
        "nbc2": (NBC2(dim_input=4, dim_output=4, n_layers=2, dim_hidden=48, dim_ffn=64, num_freqs=5, block_kwargs=bk), torch.randn(2, 5, 11, 4)),
        "nbc": (NBC(dim_input=4, dim_output=4, n_layers=2, encoder_kernel_size=4, n_heads=2, hidden_size=48, ffn_size=64), torch.randn(2, 3, 13, 4)),
        "blstm": (BLSTM2_FC1(dim_input=4, dim_output=4, hidden_size=(128, 128)), torch.randn(1, 3, 6, 4)),
    }
    out = {}
    for name, (net, x) in cases.items():
        net.eval()  # (NBC's blocks carry dropout 0.1: eval mode makes the fixture deterministic; none of the three has mode-dependent statistics)
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
                if name == "blstm":
                    p.copy_(p.half().float())
        sd32 = {k: v.clone() for k, v in net.state_dict().items()}
        net = net.double()
        y = net(x.double())
        r = torch.randn(y.shape)
        (y * r.double()).sum().backward()
        out[f"{name}/x"], out[f"{name}/y"], out[f"{name}/r"] = x.numpy(), y.detach().float().numpy(), r.numpy()
        for k, v in sd32.items():
            if k.endswith("rel_pos.pe"):
                continue
            out[f"{name}/param/{k}"] = v.numpy().astype(np.float16) if name == "blstm" else v.numpy()
        for k, p in net.named_parameters():
            g = p.grad.float().numpy()
            if name == "blstm" and g.ndim == 2 and g.shape[0] >= 512:
                g = g[::8]
            out[f"{name}/grad/{k}"] = g
    np.savez_compressed(HERE / "nb_models_native.npz", **out)
    print("written: nb_models_native.npz", sum(v.nbytes for v in out.values()) // 1024, "KB uncompressed")


if __name__ == "__main__":
    main()
