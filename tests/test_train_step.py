"""Whole training step on the HIP path vs autograd through the fp64 oracle: loss, separated signals and the
gradient of EVERY parameter (fp32 stream), then one clip+Adam update; plus bf16-stream sanity."""
import pytest
import torch

from nbss_amd._lib import NBSS_BF16, NBSS_F32
from nbss_amd.engine import SpatialNetEngine, TrainStep
from oracle import io_ref
from oracle import spatialnet_ref as ref
from util import rel_l2


def make(backend, dtype, B, C, S, N, L, seed=0, num_freqs=129):
    eng = SpatialNetEngine(backend.lib, backend.device, dim_input=2 * C, dim_output=2 * S, num_freqs=num_freqs, num_layers=L, dtype=dtype)
    p = ref.init_params(num_layers=L, num_freqs=num_freqs, dim_input=2 * C, dim_output=2 * S, seed=seed)
    eng.load_params(p)
    g = torch.Generator().manual_seed(seed + 1)
    src = torch.randn(B, S, N, generator=g)
    mix = torch.stack([src.sum(1) * (0.5 + 0.1 * c) + 0.05 * torch.randn(B, N, generator=g) for c in range(C)], 1)
    return eng, p, mix, src


def oracle_step(p, mix, src, L, n_fft=256):
    p64 = {}
    seen = {}
    for k, v in p.items():
        if id(v) not in seen:
            seen[id(v)] = v.double().clone().requires_grad_(True)
        p64[k] = seen[id(v)]
    loss, yr_hat, out = io_ref.train_forward(mix.double(), src.double(), p64, L, n_fft=n_fft, hop=n_fft // 2)
    loss.backward()
    return loss.detach(), yr_hat.detach(), {k: v.grad for k, v in p64.items()}


@pytest.mark.parametrize("L", [1, 2])
def test_train_step_fp32_matches_autograd(backend, L):
    if backend.name == "emu" and L > 1:
        pytest.skip("emulator: one layer is enough to exercise the sequencing (the gpu run covers L=2)")
    B, C, S, N = (1, 2, 2, 768) if backend.name == "emu" else (2, 6, 2, 8000)
    eng, p, mix, src = make(backend, NBSS_F32, B, C, S, N, L)
    ts = TrainStep(eng)
    x, yr = mix.to(backend.device), src.to(backend.device)
    loss, yr_hat, dout, xin, _ = ts.forward_loss(x, yr)
    eng.backward(xin, dout)
    wl, wy, wg = oracle_step(p, mix, src, L)
    assert abs(float(loss) - float(wl)) < 1e-3 * max(1.0, abs(float(wl)))
    assert rel_l2(yr_hat, wy) < 1e-3
    views = eng.param_views(eng.grads)
    bad = []
    for k, g in wg.items():
        err = rel_l2(views[k], g)
        if err > 2e-3:
            bad.append((k, err))
    assert not bad, bad
    # one optimizer step == clip_grad_norm_(5) + Adam(lr=1e-3) on the oracle gradients
    flat_g = eng.grads.clone()
    ts.step_count = 0
    from nbss_amd import ops
    ops.clip_adam_step(eng.lib, eng.params, eng.grads, ts.m, ts.v, ts.scratch, 1, lr=1e-3, max_norm=5.0)
    ref_p = torch.cat([v.reshape(-1) for v in [eng.param_views(torch.zeros_like(flat_g))[k] for k in eng.table]])  # shape only
    norm = float(flat_g.norm())
    assert abs(float(ts.scratch[0]) - norm) < 1e-4 * norm
    assert float(eng.grads.abs().max()) == 0.0


def test_train_step_bf16_runs_and_learns(backend):
    if backend.name == "emu":
        pytest.skip("multi-step bf16 training is a gpu test (minutes on the emulator)")
    B, C, S, N, L = (2, 6, 2, 8000, 2)
    eng, p, mix, src = make(backend, NBSS_BF16, B, C, S, N, L)
    ts = TrainStep(eng, lr=1e-3)
    x, yr = mix.to(backend.device), src.to(backend.device)
    wl, _, _ = oracle_step(p, mix, src, L)
    l0 = float(ts.step(x, yr))
    assert abs(l0 - float(wl)) < 0.05 * max(1.0, abs(float(wl)))  # bf16 stream vs fp64 oracle
    losses = [l0] + [float(ts.step(x, yr)) for _ in range(3 if backend.name == "emu" else 10)]
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]  # same batch repeatedly: the loss must go down


def test_16khz_geometry(backend):
    """n_fft 512 / hop 256 -> 257 frequencies (the reference's 16-kHz setting, configs/SpatialNet.yaml:25 `num_freqs: 129 / 257`):
    fp32-stream forward (loss, separated signals) and one bf16-stream training step with every parameter gradient vs the fp64 oracle."""
    B, C, S, N, L = (1, 2, 2, 1536, 1) if backend.name == "emu" else (2, 6, 2, 16000, 2)
    eng, p, mix, src = make(backend, NBSS_F32, B, C, S, N, L, num_freqs=257)
    ts = TrainStep(eng, n_fft=512)
    x, yr = mix.to(backend.device), src.to(backend.device)
    wl, wy, wg = oracle_step(p, mix, src, L, n_fft=512)
    loss, yr_hat, _, _, _ = ts.forward_loss(x, yr, need_grad=False)
    assert abs(float(loss) - float(wl)) < 1e-3 * max(1.0, abs(float(wl)))
    assert rel_l2(yr_hat, wy) < 1e-3
    eng, p, mix, src = make(backend, NBSS_BF16, B, C, S, N, L, num_freqs=257)
    ts = TrainStep(eng, n_fft=512)
    loss, yr_hat, dout, xin, _ = ts.forward_loss(x, yr)
    eng.backward(xin, dout)
    assert abs(float(loss) - float(wl)) < 0.05 * max(1.0, abs(float(wl)))
    views = eng.param_views(eng.grads)
    bad = [(k, rel_l2(views[k], g)) for k, g in wg.items() if rel_l2(views[k], g) > 0.12]  # bf16 stream, whole network (tests/test_e2e_headline.py)
    assert not bad, bad


def test_graft_smoke_body_on_the_emulator(emu_lib):
    """__graft_entry__.smoke() runs this body on cuda:0 at the round's end; here the same code on the host emulator at a small grid"""
    import __graft_entry__ as g
    r = g._smoke(emu_lib, torch.device("cpu"), F=9, T=21, L=2)
    assert r["forward"] < 3e-2 and r["walk_forward"] < 3e-2 and r["gradient"] < 6e-2, r
