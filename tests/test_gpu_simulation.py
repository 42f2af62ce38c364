"""data_loaders/gpu_simulation.py: the batched torch pipeline against scipy / the reference's numpy functions
(on host tensors; the same code runs on the HIP device, checked by the gpu test) and the loader contract."""
from pathlib import Path

import numpy as np
import pytest
import torch

from data_loaders import gpu_simulation as gs


def test_fft_convolve_and_alignment_match_scipy():
    from scipy.signal import fftconvolve
    g = torch.Generator().manual_seed(0)
    wav, rir = torch.randn(2, 2, 700, generator=g, dtype=torch.float64), torch.randn(2, 2, 3, 90, generator=g, dtype=torch.float64) * 0.1
    rir[:, :, :, 17] = 1.0
    rir[:, :, 0, 13] = 2.0  # direct path of the reference channel at sample 13
    rv, tg = gs.convolve_aligned(wav, rir)
    for b in range(2):
        for s in range(2):
            want = fftconvolve(wav[b, s].numpy()[None], rir[b, s].numpy(), mode="full", axes=-1)[:, 13:13 + 700]  # mix.py:122-134
            assert np.abs(rv[b, s].numpy() - want).max() < 1e-9 and torch.equal(rv, tg)


def test_sir_snr_and_peak_scaling():
    dm = gs.SimulatedRoomDataModule(batch_size=[3, 3], num_samples=[6, 3, 3], audio_time_len=[0.5, 0.5, 0.5], device="cpu")
    g = torch.Generator().manual_seed(1)
    src = torch.randn(3, 2, 4000, generator=g)
    sir, snr = torch.tensor([-5.0, 0.0, 4.0]), torch.tensor([3.0, 10.0, 20.0])
    mix, tgt, paras = gs.mix_batch(src, dm._rirs(3, g), dm.Cs, sir, snr, g)
    assert mix.shape == (3, 6, 4000) and tgt.shape == (3, 2, 6, 4000)
    e = tgt.pow(2).sum((2, 3))
    assert torch.allclose(10 * torch.log10(e[:, 0] / e[:, 1]), sir, atol=1e-3)  # targets are the reverberant images here
    assert torch.allclose(paras["snr"], snr, atol=1e-3)
    assert torch.allclose(torch.maximum(mix.abs().amax((1, 2)), tgt.abs().amax((1, 2, 3))), torch.full((3,), 0.9), atol=1e-5)


def test_diffuse_noise_coherence_and_reference():
    pos = torch.tensor([[0.0, 0, 0], [0.05, 0, 0], [0.2, 0, 0]])
    dsc, Cs = gs.diffuse_mixing_matrices(pos, 8000)
    recon = torch.einsum("fmi,fmj->fij", Cs.conj(), Cs).real  # C^H C = coherence
    assert (recon[1:] - dsc.permute(2, 0, 1)[1:]).abs().max() < 1e-9
    g = torch.Generator().manual_seed(2)
    x = gs.gen_diffuse_noise(torch.randn(3, 160000, generator=g, dtype=torch.float64), 160000, Cs)
    X = gs._stft_scipy(x, 256)
    psd = (X.abs() ** 2).mean(-1)
    coh01 = ((X[0] * X[1].conj()).mean(-1) / torch.sqrt(psd[0] * psd[1])).real
    k = torch.arange(4, 60)
    assert (coh01[k] - dsc[0, 1, k]).abs().mean() < 0.05  # measured coherence of the close pair follows sinc(w d / c)
    ref = Path("/root/reference/data_loaders/utils/diffuse_noise.py")
    if ref.exists():  # the reference's own generator on the same noise and matrices (it cuts a random start: use exact-length noise)
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_diffuse_noise", ref)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        noise = torch.randn(3 * 4000, generator=g, dtype=torch.float64)
        want = mod.gen_diffuse_noise(noise.numpy(), 4000, Cs.numpy(), nfft=256, rng=np.random.default_rng(0))
        got = gs.gen_diffuse_noise(noise.reshape(3, 4000), 4000, Cs)
        assert np.abs(got.numpy() - want).max() < 1e-8 * max(1.0, np.abs(want).max())
        dsc_ref, Cs_ref = mod.gen_desired_spatial_coherence(pos.numpy(), 8000)
        assert np.abs(dsc_ref - dsc.numpy()).max() < 1e-12


def test_loader_contract_and_determinism():
    dm = gs.SimulatedRoomDataModule(batch_size=[2, 2], num_samples=[4, 2, 2], audio_time_len=[0.25, 0.25, 0.25], device="cpu")
    a = list(dm.batches(0, 0, 1, 0))
    b = list(dm.batches(0, 0, 1, 0))
    assert len(a) == 2 and a[0][0].shape == (2, 6, 2000) and a[0][1].shape == (2, 2, 6, 2000) and len(a[0][2]) == 2
    assert torch.equal(a[0][0], b[0][0]) and not torch.equal(a[0][0], a[1][0])
    r0, r1 = list(dm.batches(0, 0, 2, 0)), list(dm.batches(0, 1, 2, 0))
    assert {p["index"] for _, _, ps in r0 for p in ps}.isdisjoint({p["index"] for _, _, ps in r1 for p in ps})


@pytest.mark.gpu
def test_simulation_on_device_feeds_training_step():
    dm = gs.SimulatedRoomDataModule(batch_size=[2, 2], num_samples=[4, 2, 2], audio_time_len=[1.0, 1.0, 1.0], device="cuda:0")
    x, ys, paras = next(iter(dm.batches(0)))
    assert x.is_cuda and x.shape == (2, 6, 8000) and torch.isfinite(x).all() and abs(float(x.abs().max()) - 0.9) < 0.2
    host = gs.SimulatedRoomDataModule(batch_size=[2, 2], num_samples=[4, 2, 2], audio_time_len=[1.0, 1.0, 1.0], device="cpu")
    assert host.Cs.shape == dm.Cs.shape
    # numbers, not only shapes: the SAME sources, RIRs, SIR / SNR draws and sensor noises through the device pipeline (fp32, rocFFT) and through the
    # host pipeline in fp64 — which the CPU tests above pin to scipy.signal.fftconvolve and the reference's numpy functions (mix.py:122-134,269-303,
    # diffuse_noise.py:64-93)
    g = torch.Generator().manual_seed(5)
    src = torch.randn(3, 2, 8000, generator=g, dtype=torch.float64)
    rir = host._rirs(3, g).double()
    sir, snr = torch.tensor([-5.0, 0.0, 4.0], dtype=torch.float64), torch.tensor([3.0, 10.0, 20.0], dtype=torch.float64)
    white = torch.randn(3, 6, 8000, generator=g, dtype=torch.float64)
    Cs64 = gs.diffuse_mixing_matrices(host.pos_mics, 8000)[1]
    assert float((dm.Cs.cpu().to(torch.complex128) - Cs64).abs().max()) < 1e-6  # the mixing matrices the device module built
    want_mix, want_tgt, want_p = gs.mix_batch(src, rir, Cs64, sir, snr, None, white=white)
    d = lambda t: t.float().to("cuda:0")  # noqa: E731
    mix, tgt, p = gs.mix_batch(d(src), d(rir), dm.Cs, d(sir), d(snr), None, white=d(white))
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())  # noqa: E731
    assert mix.is_cuda and rel(mix, want_mix) < 1e-4 and rel(tgt, want_tgt) < 1e-4, (rel(mix, want_mix), rel(tgt, want_tgt))
    assert torch.allclose(p["snr"].double().cpu(), want_p["snr"], atol=1e-3) and torch.allclose(p["scale"].double().cpu(), want_p["scale"], rtol=1e-4)
