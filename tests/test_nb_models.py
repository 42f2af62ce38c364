"""The narrow-band model family behind the shared trainer (BASELINE configs 1 and 4): drop-in
models.arch.{blstm2_fc1, NBC2, NBC, NBSS} in plain PyTorch.  Pinned against fixtures produced BY THE REFERENCE'S OWN MODULES
(tests/golden/make_golden.py: nb_models): strict state_dict interchange, forward output and every parameter gradient; the host
(torch) paths of models.io.{stft,loss} against the oracle; and the `trainer.accelerator=cpu` fit of BASELINE config 1."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import io_ref
from util import rel_l2

ROOT = Path(__file__).resolve().parent.parent
Z = np.load(ROOT / "tests" / "golden" / "nb_models_tiny.npz")


def _build(name):
    from models.arch.blstm2_fc1 import BLSTM2_FC1
    from models.arch.NBC import NBC
    from models.arch.NBC2 import NBC2
    from models.arch.NBSS import NBSS
    bk = {"n_heads": 2, "dropout": 0, "conv_kernel_size": 3, "n_conv_groups": 4, "norms": ("LN", "GBN", "GBN"),
          "group_batch_norm_kwargs": {"share_along_sequence_dim": False}}
    return {
        "blstm": lambda: BLSTM2_FC1(dim_input=4, dim_output=4, hidden_size=(8, 6)),
        "nbc2": lambda: NBC2(dim_input=4, dim_output=4, n_layers=2, dim_hidden=16, dim_ffn=32, num_freqs=5, block_kwargs=bk),
        "nbc": lambda: NBC(dim_input=4, dim_output=4, n_layers=2, encoder_kernel_size=4, n_heads=4, hidden_size=16, ffn_size=32),
        "nbss": lambda: NBSS(n_channel=2, n_speaker=2, n_fft=64, n_overlap=32, ref_channel=1, arch="NB_BLSTM", arch_kwargs={"hidden_size": (8, 6)}),
    }[name]()


@pytest.mark.parametrize("name", ["blstm", "nbc2", "nbc", "nbss"])
def test_matches_reference_fixture(name):
    torch.manual_seed(123)  # different init: everything must come from the fixture
    net = _build(name).eval()  # (the fixtures were made in eval mode: NBC's default dropout is 0.1)
    sd = {k[len(name) + 7:]: torch.from_numpy(Z[k]) for k in Z.files if k.startswith(f"{name}/param/")}
    net.load_state_dict(sd, strict=True)  # same keys and shapes as the reference module
    x, r = torch.from_numpy(Z[f"{name}/x"]), torch.from_numpy(Z[f"{name}/r"])
    y = net(x)
    assert rel_l2(y, torch.from_numpy(Z[f"{name}/y"])) < 2e-5
    (y * r).sum().backward()
    want = {k[len(name) + 6:]: torch.from_numpy(Z[k]) for k in Z.files if k.startswith(f"{name}/grad/")}
    got = dict(net.named_parameters())
    assert set(want) == set(got)
    top = max(float(g.norm()) for g in want.values())
    for k, g in want.items():  # (a few gradients are analytically zero, e.g. the key bias under a softmax: absolute floor)
        err = float((got[k].grad - g).norm())
        assert err <= 2e-4 * float(g.norm()) + 1e-6 * top, (k, err, float(g.norm()))


def test_nbc2_group_batch_norm_groups_are_utterances():
    """GroupBatchNorm shares statistics inside each group of `num_freqs` consecutive sequences (= one utterance): an utterance's
    output must not depend on the other utterances of the batch (NBC2.py:111-145)"""
    net = _build("nbc2").eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 9, 4, generator=g)
    with torch.no_grad():
        assert rel_l2(net(x)[1], net(x[1:2])[0]) < 1e-5


@pytest.mark.parametrize("S", [1, 2, 3])
def test_host_loss_matches_oracle(S):
    """models.io.loss on host tensors (torch closed forms) == the oracle's restatement of torchmetrics' si_sdr / pit"""
    from models.io.loss import Loss, neg_si_sdr
    g = torch.Generator().manual_seed(S)
    t = torch.randn(4, S, 900, generator=g)
    p = (t[:, torch.randperm(S, generator=g)] + 0.3 * torch.randn(4, S, 900, generator=g)).requires_grad_(True)
    loss, perms, _ = Loss(neg_si_sdr, pit=True)(yr_hat=p, yr=t, reorder=False, reduce_batch=True)
    wl, items, wperm = io_ref.pit_neg_si_sdr(p.detach().double(), t.double())
    assert abs(float(loss) - float(wl)) < 1e-4 and torch.equal(perms, wperm)
    per_item, _, _ = Loss(neg_si_sdr, pit=True)(yr_hat=p, yr=t, reduce_batch=False)
    assert rel_l2(per_item, items) < 1e-5
    assert rel_l2(neg_si_sdr(p.detach(), t), io_ref.neg_si_sdr(p.detach().double(), t.double())) < 1e-5
    loss.backward()
    p64 = p.detach().double().requires_grad_(True)
    io_ref.pit_neg_si_sdr(p64, t.double())[0].backward()
    assert rel_l2(p.grad, p64.grad) < 1e-4


def test_train_module_host_forward_is_the_reference_sequence():
    """TrainModule.forward on host tensors = stft -> Norm('frequency') -> arch -> inorm -> istft (SharedTrainer.py:104-132)"""
    from SharedTrainer import TrainModule
    from models.arch.blstm2_fc1 import BLSTM2_FC1
    from models.io.loss import Loss, neg_si_sdr
    from models.io.norm import Norm
    from models.io.stft import STFT
    torch.manual_seed(0)
    arch = BLSTM2_FC1(dim_input=4, dim_output=4, hidden_size=(8, 6))
    m = TrainModule(arch, channels=[0, 2], ref_channel=2, stft=STFT(256, 128), norm=Norm("frequency"), loss=Loss(neg_si_sdr, pit=True))
    x = torch.randn(2, 3, 2000)
    yr_hat, _ = m(x)
    X = io_ref.stft(x[:, [0, 2]])
    Xn, XrMM = io_ref.norm_frequency_online(X, 1)
    out = arch(io_ref.to_real_layout(Xn))
    want = io_ref.istft(io_ref.from_real_layout(out) * XrMM, 2000)
    assert rel_l2(yr_hat, want) < 1e-5
    with pytest.raises(RuntimeError, match="HIP"):  # SpatialNet has no host path
        from models.arch.SpatialNet import SpatialNet
        SpatialNet(dim_input=4, dim_output=4, dim_squeeze=8, num_layers=1, num_freqs=129, dim_hidden=96, dim_ffn=192, num_heads=4)(torch.randn(1, 129, 8, 4))


def test_fit_nb_blstm_on_cpu():
    """BASELINE config 1: NB-BLSTM, 2 speakers, 2 channels, n_fft 256 (129 freqs), 1-s utterances, `SharedTrainer fit` with
    trainer.accelerator=cpu"""
    from SharedTrainer import TrainCLI
    cli = TrainCLI(argv=["fit", "--config", str(ROOT / "configs" / "NB-BLSTM.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml"),
                         "--model.arch.dim_input=4", "--model.arch.dim_output=4", "--model.channels=[0,1]", "--trainer.accelerator=cpu",
                         "--data.audio_time_len=[1.0,1.0,1.0]", "--data.num_samples=[8,2,2]", "--trainer.max_epochs=2", "--model.arch.hidden_size=[32,16]"])
    log = cli.result["log"]
    assert len(log) == 2 and log[0]["device"] == "cpu"
    assert all(np.isfinite(r["train/neg_si_sdr"]) and np.isfinite(r["val/neg_si_sdr"]) for r in log)
    assert log[1]["train/neg_si_sdr"] < log[0]["train/neg_si_sdr"]


@pytest.mark.parametrize("cfg", ["NB-BLSTM.yaml", "NBC2.yaml"])
def test_reference_configs_build(cfg):
    """the reference's own YAML files (when the tree is present) and this repo's copies instantiate the drop-in modules"""
    from SharedTrainer import build_module, parse_cli
    for path in (ROOT / "configs" / cfg, Path("/root/reference/configs") / cfg):
        if not path.exists():
            continue
        _, c = parse_cli(["fit", "--config", str(path), "--model.arch.dim_input=12", "--model.arch.dim_output=4"])
        m = build_module(c)
        assert type(m.arch).__module__ in ("models.arch.blstm2_fc1", "models.arch.NBC2")
        with torch.no_grad():
            y, _ = m(torch.randn(1, 6, 2048))
        assert y.shape == (1, 2, 2048)


@pytest.mark.gpu
def test_group_norm_gradients_beyond_128_samples_on_the_device():
    """torch 2.10+rocm7.0: torch.nn.functional.group_norm's backward returns wrong weight / bias gradients on the HIP device once the batch exceeds
    128 samples (tests/diag/torch_group_norm_check.py) — models.arch.base.norm.group_norm, which every GroupNorm of the torch.nn archs here goes
    through, must give the CPU result at the batch sizes training uses (one sample per (batch, frequency) sequence: 258 at batch 2)"""
    from models.arch.base.norm import GroupNorm
    torch.manual_seed(0)
    for N in (64, 129, 1032):
        h = torch.randn(N, 48, 16)
        r = torch.randn_like(h)
        grads = []
        for dev in ("cpu", "cuda"):
            gn = GroupNorm(seq_last=True, num_groups=8, num_channels=48)
            with torch.no_grad():
                gn.weight.copy_(torch.linspace(0.5, 1.5, 48))
                gn.bias.copy_(torch.linspace(-0.2, 0.2, 48))
            gn = gn.to(dev)
            x = h.detach().clone().to(dev).requires_grad_(True)
            (gn(x) * r.to(dev)).sum().backward()
            grads.append([gn.weight.grad.cpu(), gn.bias.grad.cpu(), x.grad.cpu()])
        for a, b in zip(*grads):
            assert float((a - b).norm() / b.norm()) < 1e-5, N


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["NB_BLSTM", "NBC2"])
def test_nbss_on_the_device_uses_the_hip_stft_and_equals_the_host_path(hip_lib, arch):
    """NBSS.forward on a HIP tensor (n_fft 256 / hop 128: BASELINE config 1's geometry) runs the STFT / iSTFT kernels of signal.hip — and NBC2 its native
    path — and returns what the host path (torch.stft / torch.istft / torch.nn, pinned to the reference fixture above) returns; gradients too; the
    state_dict carries no key the reference's NBSS lacks"""
    from models.arch.NBSS import NBSS
    torch.manual_seed(4)
    kw = {"hidden_size": (16, 8)} if arch == "NB_BLSTM" else {"n_layers": 1, "dim_hidden": 96, "dim_ffn": 192, "num_freqs": 129}
    net = NBSS(n_channel=2, n_speaker=2, n_fft=256, n_overlap=128, ref_channel=0, arch=arch, arch_kwargs=kw)
    assert not any(k.startswith("_stft") or k.startswith("stft") for k in net.state_dict())
    x = torch.randn(2, 2, 8000)
    r = torch.randn(2, 2, 8000)
    y = net(x)
    (y * r).sum().backward()
    want = {k: p.grad.clone() for k, p in net.named_parameters()}
    net.zero_grad()
    dev = net.cuda()
    yd = dev(x.cuda())
    assert dev._io().hip_ok and yd.is_cuda
    assert rel_l2(yd, y) < 2e-4
    (yd * r.cuda()).sum().backward()
    top = max(float(g.norm()) for g in want.values())
    for k, p in dev.named_parameters():
        assert float((p.grad.cpu() - want[k]).norm()) <= 2e-3 * float(want[k].norm()) + 1e-5 * top, k
