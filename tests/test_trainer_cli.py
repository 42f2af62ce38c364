"""The LightningCLI-shaped surface: the reference's own configs/SpatialNet.yaml (when present) and this repo's configs
parse with the reference's command line and build the drop-in modules; `fit` itself is a gpu test."""
from pathlib import Path

import pytest
import torch

from SharedTrainer import TrainCLI, build_module, parse_cli

ROOT = Path(__file__).resolve().parent.parent
ARGS = ["--model.arch.dim_input=12", "--model.arch.dim_output=4", "--model.arch.num_freqs=129", "--trainer.precision=bf16-mixed",
        "--model.exp_name", "notag", "--data.batch_size=[2,4]"]


@pytest.mark.parametrize("cfg", [ROOT / "configs" / "SpatialNet.yaml", Path("/root/reference/configs/SpatialNet.yaml")])
def test_config_builds_dropin_modules(cfg):
    if not cfg.exists():
        pytest.skip("reference tree not present")
    sub, c = parse_cli(["fit", "--config", str(cfg), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml")] + ARGS)
    assert sub == "fit" and c["trainer"]["precision"] == "bf16-mixed" and c["trainer"]["gradient_clip_val"] == 5
    assert c["data"]["init_args"]["batch_size"] == [2, 4]
    m = build_module(c)
    assert type(m.arch).__module__ == "models.arch.SpatialNet"
    assert sum(p.numel() for p in m.arch.parameters()) == 1191092  # reference: 1.2 M (images/model_size_and_flops.png)
    sd = m.arch.state_dict()
    for k, shape in {"encoder.weight": (96, 12, 5), "layers.3.fconv1.1.weight": (96, 12, 5), "layers.7.full.weight": (8, 129, 129),
                     "layers.0.mhsa.in_proj_weight": (288, 96), "layers.5.tconvffn.6.weight": (192,), "layers.2.tconvffn.8.weight": (192, 24, 3),
                     "decoder.weight": (4, 96)}.items():
        assert tuple(sd[k].shape) == shape, k  # checkpoint contract, SURVEY.md §8(b)
    assert sd["layers.7.full.weight"].data_ptr() == sd["layers.0.full.weight"].data_ptr()  # full_share=0: one shared LinearGroup
    assert m.loss.name == "neg_si_sdr" and m.loss.pit and m.optimizer == ("Adam", {"lr": 0.001})


@pytest.mark.gpu
def test_fit_one_epoch_on_gpu():
    argv = ["fit", "--config", str(ROOT / "configs" / "SpatialNet.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml"),
            "--model.arch.dim_input=12", "--model.arch.dim_output=4", "--trainer.precision=bf16-mixed", "--trainer.max_epochs=2",
            "--data.num_samples=[8,2,2]", "--data.audio_time_len=[1.0,1.0,1.0]", "--model.arch.num_layers=2"]
    cli = TrainCLI(argv=argv)
    log = cli.result["log"]
    assert len(log) == 2 and all(torch.isfinite(torch.tensor(r["train/neg_si_sdr"])) for r in log)
    assert log[1]["train/neg_si_sdr"] < log[0]["train/neg_si_sdr"]
    assert all(torch.isfinite(torch.tensor(r["val/neg_si_sdr"])) for r in log)  # the epoch's validation pass (val_metric: loss)
    # a scheduler on the validation metric: the logged learning rates are what torch's ReduceLROnPlateau makes of the logged validation losses
    cli = TrainCLI(argv=argv + ["--model.lr_scheduler=[ReduceLROnPlateau, {factor: 0.5, patience: 0, threshold: 0.5}]", "--trainer.max_epochs=3"])
    log = cli.result["log"]
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=0, threshold=0.5)
    for r in log:
        ref.step(r["val/neg_si_sdr"])
        assert abs(r["lr"] - opt.param_groups[0]["lr"]) < 1e-12, log


def test_plateau_rule_matches_torch():
    from SharedTrainer import _Plateau
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=1, threshold=1e-2, cooldown=1, min_lr=0.05)
    mine, lr = _Plateau(mode="min", factor=0.5, patience=1, threshold=1e-2, cooldown=1, min_lr=0.05), 1.0
    for m in [5.0, 4.0, 4.0, 4.0, 3.99, 4.1, 4.2, 4.3, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]:
        ref.step(m)
        lr = mine.step(m, lr)
        assert abs(lr - opt.param_groups[0]["lr"]) < 1e-12, (m, lr, opt.param_groups[0]["lr"])


def test_plateau_fuzz_and_state_round_trip_with_torch():
    """random metric walks, every option (threshold_mode abs / rel, max mode, eps, cooldown, min_lr): the same learning rates as torch's
    ReduceLROnPlateau; the state_dict loads INTO torch's scheduler (what a Lightning resume does) and torch's state loads into ours"""
    import random
    from SharedTrainer import _Plateau
    rng = random.Random(0)
    for trial in range(40):
        kw = dict(mode=rng.choice(["min", "max"]), factor=rng.choice([0.1, 0.5, 0.9]), patience=rng.randint(0, 3), threshold=rng.choice([1e-4, 1e-2, 0.3]),
                  threshold_mode=rng.choice(["rel", "abs"]), cooldown=rng.randint(0, 2), min_lr=rng.choice([0.0, 1e-9, 1e-3]), eps=rng.choice([1e-8, 1e-3]))
        lr0 = rng.choice([1.0, 1e-3, 3e-8])
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=lr0)
        ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, **kw)
        mine, lr = _Plateau(**kw), lr0
        metrics = [rng.choice([1.0, 2.0, 0.5, 1.001, 0.999]) * rng.random() for _ in range(30)]
        for i, m in enumerate(metrics):
            ref.step(m)
            lr = mine.step(m, lr)
            assert lr == opt.param_groups[0]["lr"], (trial, i, kw, lr, opt.param_groups[0]["lr"])
            if i == 14:  # resume both ways mid-run
                opt2 = torch.optim.SGD([p], lr=lr)
                ref2 = torch.optim.lr_scheduler.ReduceLROnPlateau(opt2, **kw)
                ref2.load_state_dict(mine.state_dict(lr))
                mine2 = _Plateau(**kw)
                mine2.load_state_dict(ref.state_dict())
                assert (ref2.best, ref2.num_bad_epochs, ref2.cooldown_counter) == (ref.best, ref.num_bad_epochs, ref.cooldown_counter)
                ref, opt, mine = ref2, opt2, mine2
    with pytest.raises(NotImplementedError):
        _Plateau(mode="min", monitor="x")


def test_streamer_cache_follows_the_weights():
    """forward_streaming caches its streamer (packed weight copies, a captured graph): an in-place weight update must rebuild it"""
    from SharedTrainer import TrainModule
    from models.arch.OnlineSpatialNet import OnlineSpatialNet
    torch.manual_seed(0)
    arch = OnlineSpatialNet(dim_input=4, dim_output=4, num_layers=1, dim_squeeze=8, num_freqs=9, encoder_kernel_size=5, dim_hidden=32, dim_ffn=64, num_heads=4,
                            dropout=(0, 0, 0), kernel_size=(5, 3), conv_groups=(8, 8), norms=["LN", "LN", "GN", "LN", "LN", "LN"], full_share=0,
                            attention="ret(2)", decay=[4, 5, 9, 10], rope=False).eval()
    from models.io.norm import Norm
    from models.io.stft import STFT
    from models.io.loss import Loss
    m = TrainModule(arch=arch, channels=[0, 1], ref_channel=0, stft=STFT(n_fft=16, n_hop=8), norm=Norm(mode="frequency", online=True),
                    loss=Loss(loss_func="models.io.loss.neg_si_sdr", pit=True))
    x = torch.randn(1, 2, 400)
    y0, _ = m.forward_streaming(x, chunk=4, use_graph=False)
    k0 = m._streamer_key
    y1, _ = m.forward_streaming(x, chunk=4, use_graph=False)
    assert m._streamer_key == k0 and torch.equal(y0, y1)  # unchanged weights: same streamer
    with torch.no_grad():
        for p in arch.parameters():
            p.mul_(1.5)
    y2, _ = m.forward_streaming(x, chunk=4, use_graph=False)
    assert m._streamer_key != k0
    want = m.forward(x)[0]
    assert torch.allclose(y2, want, atol=1e-4, rtol=1e-3), float((y2 - want).abs().max())


@pytest.mark.gpu
def test_validate_test_predict_on_gpu(tmp_path):
    """the forward-only path behind the reference's validate / test / predict subcommands"""
    base = ["--config", str(ROOT / "configs" / "SpatialNet.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml"),
            "--model.arch.dim_input=12", "--model.arch.dim_output=4", "--trainer.precision=bf16-mixed", "--data.num_samples=[8,4,4]",
            "--data.audio_time_len=[1.0,1.0,1.0]", "--model.arch.num_layers=2"]
    for sub, key in (("validate", "val"), ("test", "test")):
        rec = TrainCLI(argv=[sub] + base).result
        assert rec["batches"] >= 1 and all(torch.isfinite(torch.tensor(v)) for v in rec.values())
        assert f"{key}/neg_si_sdr" in rec and f"{key}/si_sdr_improvement_dB" in rec
    out = TrainCLI(argv=["predict"] + base + [f"--trainer.default_root_dir={tmp_path}"]).result["yr_hat"]
    assert len(out) >= 1 and out[0].shape[1:] == (2, 8000) and torch.isfinite(out[0]).all()
    assert (tmp_path / "predict_00000.pt").exists()


@pytest.mark.gpu
def test_large_and_16khz_through_the_cli(tmp_path):
    """SpatialNet-large ("for large" comments of configs/SpatialNet.yaml) at 16 kHz (n_fft 512 -> 257 bins) through
    validate / predict / fit (generic backward, csrc/gbwd.hip); the small model trains at 16 kHz"""
    base = ["--config", str(ROOT / "configs" / "SpatialNet.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml"),
            "--model.arch.dim_input=12", "--model.arch.dim_output=4", "--model.arch.num_freqs=257", "--model.stft.n_fft=512", "--model.stft.n_hop=256",
            "--data.sample_rate=16000", "--trainer.precision=bf16-mixed", "--data.num_samples=[4,2,2]", "--data.audio_time_len=[1.0,1.0,1.0]"]
    large = ["--model.arch.num_layers=3", "--model.arch.dim_hidden=192", "--model.arch.dim_ffn=384", "--model.arch.dim_squeeze=16"]
    rec = TrainCLI(argv=["validate"] + base + large).result
    assert rec["batches"] >= 1 and all(torch.isfinite(torch.tensor(v)) for v in rec.values())
    out = TrainCLI(argv=["predict"] + base + large + [f"--trainer.default_root_dir={tmp_path}"]).result["yr_hat"]
    assert out[0].shape[1:] == (2, 16000) and torch.isfinite(out[0]).all()
    log = TrainCLI(argv=["fit"] + base + large + ["--trainer.max_epochs=2"]).result["log"]
    assert len(log) == 2 and log[1]["train/neg_si_sdr"] < log[0]["train/neg_si_sdr"]
    log = TrainCLI(argv=["fit"] + base + ["--model.arch.num_layers=2", "--trainer.max_epochs=2"]).result["log"]
    assert len(log) == 2 and log[1]["train/neg_si_sdr"] < log[0]["train/neg_si_sdr"]


def test_checkpoint_roundtrip_reference_format(tmp_path, emu_lib):
    """save_checkpoint writes what the reference's Lightning trainer reads: `state_dict` with `arch.` keys + `stft.window`
    (general_steps.py:189-199), `optimizer_states[0]` as a torch.optim.Adam state_dict sliced out of the fused optimizer's flat
    moments, `lr_schedulers`, version keys.  load_checkpoint restores weights AND Adam state from such a file (also from one written
    by torch itself), and tolerates the `_orig_mod.` prefix of compiled modules."""
    import ctypes as C
    from types import SimpleNamespace

    from SharedTrainer import load_checkpoint, save_checkpoint
    from nbss_amd._lib import make_cfg
    from nbss_amd.params import param_table
    _, c = parse_cli(["fit", "--config", str(ROOT / "configs" / "SpatialNet.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml")] + ARGS
                     + ["--model.arch.num_layers=2"])
    torch.manual_seed(1)
    m1 = build_module(c)
    n = sum(p.numel() for p in m1.arch.parameters())
    table = param_table(emu_lib, make_cfg(1, 129, 16, 12, 4, L=2))
    eng = SimpleNamespace(table=table)

    def fake_ts(m, v, step, lr):
        return SimpleNamespace(e=eng, m=m, v=v, step_count=step, lr=lr, betas=(0.9, 0.999), eps=1e-8, wd=0.0)

    ts1 = fake_ts(torch.randn(n), torch.rand(n), 7, 5e-4)
    path = str(tmp_path / "checkpoints" / "last.ckpt")
    save_checkpoint(path, m1, ts1, epoch=3, global_step=7)
    ck = torch.load(path, weights_only=False)
    assert set(ck) >= {"state_dict", "epoch", "global_step", "optimizer_states", "lr_schedulers", "pytorch-lightning_version"}
    assert "stft.window" in ck["state_dict"] and all(k.startswith("arch.") or k == "stft.window" for k in ck["state_dict"])
    # the optimizer state is a genuine torch.optim.Adam state_dict for `module.parameters()`
    opt = torch.optim.Adam(m1.parameters(), lr=1e-3)
    opt.load_state_dict(ck["optimizer_states"][0])
    assert opt.param_groups[0]["lr"] == 5e-4
    first = next(iter(m1.parameters()))
    off = table["encoder.weight"][0]
    assert torch.equal(opt.state[first]["exp_avg"].reshape(-1), ts1.m[off:off + first.numel()]) and float(opt.state[first]["step"]) == 7
    # ... and a checkpoint whose optimizer state was written by torch (what Lightning stores) loads into the flat buffers
    ck["optimizer_states"] = [opt.state_dict()]
    ck["state_dict"] = {k.replace("arch.", "arch._orig_mod.", 1): v for k, v in ck["state_dict"].items()}
    torch.save(ck, path)
    torch.manual_seed(2)
    m2 = build_module(c)
    ts2 = fake_ts(torch.zeros(n), torch.zeros(n), 0, 1e-3)
    assert load_checkpoint(path, m2, ts2) == 3
    for (k1, v1), (k2, v2) in zip(m1.arch.state_dict().items(), m2.arch.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert torch.equal(ts2.m, ts1.m) and torch.equal(ts2.v, ts1.v) and ts2.step_count == 7 and ts2.lr == 5e-4
    # weights-only checkpoint: loads, says that the optimizer starts fresh
    torch.save({"state_dict": ck["state_dict"], "epoch": 1}, path)
    ts3 = fake_ts(torch.zeros(n), torch.zeros(n), 0, 1e-3)
    assert load_checkpoint(path, m2, ts3) == 1 and ts3.step_count == 0


def test_unsupported_training_configs_fail_loudly():
    """the fused step hard-wires Norm('frequency', online) + uPIT: any other YAML must raise instead of training another model"""
    from SharedTrainer import _fused_step_for
    base = ["fit", "--config", str(ROOT / "configs" / "SpatialNet.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml")] + ARGS
    for extra, msg in (["--model.norm.mode=utterance"], "Norm"), (["--model.loss.pit=false"], "pit"):
        _, c = parse_cli(base + extra)
        with pytest.raises(NotImplementedError, match=msg):
            _fused_step_for(build_module(c), c, torch.device("cpu"))
    for extra, msg in (["--model.optimizer=[SGD,{lr: 0.1}]"], "optimizer"), (["--model.lr_scheduler=[StepLR,{step_size: 1}]"], "lr_scheduler"):
        _, c = parse_cli(base + extra)
        m = build_module(c)
        m.arch._engine_for = lambda dev: SimpleNamespaceEngine()  # reach the optimizer / scheduler checks without a device
        with pytest.raises(NotImplementedError, match=msg):
            _fused_step_for(m, c, torch.device("cpu"))


def test_fit_refuses_geometries_without_kernels_with_the_reason():
    from SharedTrainer import _check_train_geometry
    base = ["fit", "--config", str(ROOT / "configs" / "SpatialNet.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml")] + ARGS
    _, c = parse_cli(base + ["--model.arch.dim_hidden=128", "--model.arch.dim_ffn=256", "--model.arch.dim_squeeze=8", "--model.arch.num_layers=2"])
    with pytest.raises((NotImplementedError, Exception), match="SpatialNet-small|no HIP kernels"):
        _check_train_geometry(build_module(c))
    _, c = parse_cli(base + ["--model.arch.num_layers=2"])
    _check_train_geometry(build_module(c))  # the shipped geometry passes
    _, c = parse_cli(base + ["--model.arch.dim_hidden=192", "--model.arch.dim_ffn=384", "--model.arch.dim_squeeze=16", "--model.arch.num_layers=2"])
    _check_train_geometry(build_module(c))  # ... and so does SpatialNet-large (generic backward)


class SimpleNamespaceEngine:
    dtype = 0


def test_subcommands_fail_loudly_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("HIP device present")
    for sub in ("fit", "validate", "test", "predict"):
        with pytest.raises(RuntimeError, match="HIP kernels only"):
            TrainCLI(argv=[sub, "--config", str(ROOT / "configs" / "SpatialNet.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml"),
                           "--model.arch.dim_input=12", "--model.arch.dim_output=4"])


@pytest.mark.gpu
def test_dropin_module_autograd_on_gpu():
    """models.arch.SpatialNet.SpatialNet as a plain nn.Module under torch autograd + torch.optim (generic trainer path)"""
    from models.arch.SpatialNet import SpatialNet
    torch.manual_seed(0)
    net = SpatialNet(dim_input=12, dim_output=4, num_layers=1, dim_hidden=96, dim_ffn=192, num_heads=4, dim_squeeze=8, num_freqs=129).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = torch.randn(2, 129, 40, 12, device="cuda")
    tgt = torch.randn(2, 129, 40, 4, device="cuda")
    losses = []
    for _ in range(5):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        loss = ((y - tgt) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0]
    assert all(p.grad is not None for p in net.parameters())


# ---- the archs that are NOT the fused SpatialNet step go through the generic loop on whichever device is selected (BASELINE configs 4, 5)
NBC2_ARGS = ["--config", str(ROOT / "configs" / "NBC2.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml"),
             "--model.arch.dim_input=16", "--model.arch.dim_output=6", "--model.channels=[0,1,2,3,4,5,6,7]", "--data.num_channels=8",
             "--data.num_speakers=3", "--data.audio_time_len=[1.0,1.0,1.0]", "--data.num_samples=[4,2,2]", "--trainer.max_epochs=1"]
ONLINE_ARGS = ["--config", str(ROOT / "configs" / "onlineSpatialNet.yaml"), "--config", str(ROOT / "configs" / "datasets" / "synthetic.yaml"),
               "--model.arch.dim_input=12", "--model.arch.dim_output=4", "--model.arch.num_freqs=129", "--data.num_samples=[4,2,2]",
               "--trainer.max_epochs=1"]


def _finite(rec):
    return all(torch.isfinite(torch.tensor(float(v))) for v in rec.values() if isinstance(v, (int, float)))


def test_generic_archs_route_on_cpu(tmp_path):
    """NBC2 (8 ch -> 3 spk) and OnlineSpatialNet ret(2): fit / validate / predict through TrainCLI with trainer.accelerator=cpu — the same
    routing the gpu tests below exercise on the device; predict streams the OnlineSpatialNet chunk by chunk through OnlineStreamer"""
    small = ["--trainer.accelerator=cpu", "--model.arch.n_layers=1", "--model.arch.dim_hidden=16", "--model.arch.dim_ffn=32"]
    log = TrainCLI(argv=["fit"] + NBC2_ARGS + small).result["log"]
    assert len(log) == 1 and log[0]["device"] == "cpu" and _finite(log[0])
    rec = TrainCLI(argv=["validate"] + NBC2_ARGS + small).result
    assert rec["batches"] >= 1 and rec["device"] == "cpu" and _finite(rec)
    osmall = ["--trainer.accelerator=cpu", "--model.arch.num_layers=1", "--model.arch.dim_hidden=32", "--model.arch.dim_ffn=64",
              "--model.arch.dim_squeeze=4", "--data.audio_time_len=[0.5,0.5,1.0]"]
    log = TrainCLI(argv=["fit"] + ONLINE_ARGS + osmall + [f"--trainer.default_root_dir={tmp_path}"]).result["log"]
    assert len(log) == 1 and _finite(log[0])
    ck = str(tmp_path / "checkpoints" / "last.ckpt")
    whole = TrainCLI(argv=["predict"] + ONLINE_ARGS + osmall + ["--ckpt_path", ck, "--stream_chunk=0"]).result
    res = TrainCLI(argv=["predict"] + ONLINE_ARGS + osmall + ["--ckpt_path", ck, "--stream_chunk=8"]).result
    assert res["streamed"] and not whole["streamed"] and res["yr_hat"][0].shape == (2, 2, 8000)
    # the streamed (recurrent retention, 8 frames per step) output equals the whole-utterance (parallel retention) forward
    a, b = res["yr_hat"][0], whole["yr_hat"][0]
    assert float((a - b).norm() / b.norm()) < 5e-3


@pytest.mark.gpu
def test_nbc2_8ch_3spk_fit_validate_on_gpu():
    """BASELINE config 4: NBC2 narrow-band Conformer, 8 channels -> 3 speakers, fp32, from the shipped YAML (accelerator: gpu)"""
    log = TrainCLI(argv=["fit"] + NBC2_ARGS + ["--model.arch.n_layers=2"]).result["log"]
    assert len(log) == 1 and log[0]["device"].startswith("cuda") and _finite(log[0])
    rec = TrainCLI(argv=["validate"] + NBC2_ARGS + ["--model.arch.n_layers=2"]).result
    assert rec["batches"] >= 1 and rec["device"].startswith("cuda") and _finite(rec)


@pytest.mark.gpu
def test_online_spatialnet_fit_and_streamed_predict_on_gpu(tmp_path):
    """BASELINE config 5: OnlineSpatialNet ret(2) from the shipped YAML: one epoch of fit on the device, then `predict` on a 32-s input as
    causal chunked inference with HIP-graph-replayed steps (OnlineStreamer), equal to the whole-utterance forward"""
    args = ONLINE_ARGS + ["--model.arch.num_layers=2", "--data.audio_time_len=[1.0,1.0,32.0]", "--data.batch_size=[2,1]"]
    log = TrainCLI(argv=["fit"] + args + [f"--trainer.default_root_dir={tmp_path}"]).result["log"]
    assert len(log) == 1 and log[0]["device"].startswith("cuda") and _finite(log[0])
    ck = str(tmp_path / "checkpoints" / "last.ckpt")
    res = TrainCLI(argv=["predict"] + args + ["--ckpt_path", ck, "--stream_chunk=16"]).result
    assert res["streamed"] and res["graph_replays"] >= 32 * 8000 // 128 // 16 and res["yr_hat"][0].shape == (1, 2, 256000)
    assert res["native"]  # the shipped ret(2) geometry runs the HIP streaming kernels (csrc/online.hip), one HIP graph replay per chunk
    whole = TrainCLI(argv=["predict"] + args + ["--ckpt_path", ck, "--stream_chunk=0"]).result
    a, b = res["yr_hat"][0], whole["yr_hat"][0]
    assert torch.isfinite(a).all() and float((a - b).norm() / b.norm()) < 1e-2
