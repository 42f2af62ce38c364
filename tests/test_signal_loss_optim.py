"""STFT+norm, inorm+iSTFT (fwd/bwd), uPIT neg-SI-SDR (fwd/bwd) and clip+Adam against the oracle (torch.stft /
torch.istft / restated torchmetrics / torch.optim.Adam) on the same seeded inputs.  fp32 arithmetic everywhere:
<= 2e-5 rel-L2 (direct-DFT on the exact-f32 MFMA path vs pocketfft)."""
import pytest
import torch

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, NBSS_F32
from oracle import io_ref
from util import rel_l2

CASES = [(1, 2, 1200, 256), (2, 6, 4000, 256), (1, 3, 2300, 512)]


def cases_for(backend):
    return CASES + ([(2, 6, 32000, 256)] if backend.name == "hip" else [])


def test_stft_norm(backend):
    for (B, C, N, n_fft) in cases_for(backend):
        g = torch.Generator().manual_seed(N)
        x = torch.randn(B, C, N, generator=g)
        tab = ops.stft_tables(backend.lib, n_fft, 0, backend.device)
        X, xrmm = ops.stft_norm_fwd(backend.lib, n_fft, NBSS_F32, tab, x.to(backend.device), ref_channel=C - 1)
        Xc = io_ref.stft(x.double(), n_fft, n_fft // 2)
        Xn, mm = io_ref.norm_frequency_online(Xc, C - 1)
        assert rel_l2(xrmm, mm[:, 0]) < 2e-5
        # un-normalised spectrum: fp32 DFT accuracy; the normalised one divides by |X_ref| which can be ~1e-3, so
        # individual bins amplify the fp32 rounding of the magnitude (the fp32 reference has the same sensitivity)
        mm_l = mm[:, 0].permute(0, 1, 2)[..., None]
        assert rel_l2(X.double().cpu() * mm_l, io_ref.to_real_layout(Xc)) < 2e-5
        assert rel_l2(X, io_ref.to_real_layout(Xn)) < 1e-3
        Xb, _ = ops.stft_norm_fwd(backend.lib, n_fft, NBSS_BF16, tab, x.to(backend.device), ref_channel=C - 1)
        assert rel_l2(Xb.float(), io_ref.to_real_layout(Xn)) < 5e-3


def test_inorm_istft_fwd_bwd(backend):
    for (B, S, N, n_fft) in cases_for(backend):
        S = min(S, 3)
        g = torch.Generator().manual_seed(N + 1)
        F, T = n_fft // 2 + 1, N // (n_fft // 2) + 1
        out = torch.randn(B, F, T, 2 * S, generator=g)
        xrmm = torch.rand(B, F, T, generator=g) + 0.5
        dy = torch.randn(B, S, N, generator=g)
        tab = ops.stft_tables(backend.lib, n_fft, 0, backend.device)
        y = ops.inorm_istft_fwd(backend.lib, n_fft, tab, out.to(backend.device), xrmm.to(backend.device), N)
        o64 = out.double().requires_grad_(True)
        want = io_ref.istft(io_ref.from_real_layout(o64) * xrmm.double()[:, None], N, n_fft, n_fft // 2)
        assert rel_l2(y, want) < 2e-5
        (want * dy.double()).sum().backward()
        dout = ops.inorm_istft_bwd(backend.lib, n_fft, tab, dy.to(backend.device), xrmm.to(backend.device))
        assert rel_l2(dout, o64.grad) < 2e-5


def test_stft_istft_roundtrip(backend):
    """the reference's own smoke check (models/io/stft.py:106-112), tightened from rtol=1e-1"""
    n_fft, N = 256, 4000
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 1, N, generator=g)
    tab = ops.stft_tables(backend.lib, n_fft, 0, backend.device)
    X, xrmm = ops.stft_norm_fwd(backend.lib, n_fft, NBSS_F32, tab, x.to(backend.device), ref_channel=0)
    y = ops.inorm_istft_fwd(backend.lib, n_fft, tab, X, xrmm, N)  # (X / |X|) * |X| -> istft
    assert rel_l2(y[:, 0], x[:, 0]) < 1e-5


@pytest.mark.parametrize("S", [1, 2, 3])
def test_pit_neg_sisdr(backend, S):
    B, N = 3, 5000
    g = torch.Generator().manual_seed(S)
    t = torch.randn(B, S, N, generator=g)
    p = 0.7 * t[:, torch.randperm(S, generator=g)] + 0.5 * torch.randn(B, S, N, generator=g)
    loss, perm, dp = ops.pit_neg_sisdr(backend.lib, p.to(backend.device), t.to(backend.device))
    p64 = p.double().requires_grad_(True)
    want, _, wperm = io_ref.pit_neg_si_sdr(p64, t.double())
    want.backward()
    assert abs(float(loss) - float(want)) < 2e-5 * max(1.0, abs(float(want)))
    assert torch.equal(perm.cpu().long(), wperm)
    assert rel_l2(dp, p64.grad) < 2e-5


def test_sisdr_known_answers():
    """closed-form pins of the restated torchmetrics definitions (oracle self-check, fp64)"""
    g = torch.Generator().manual_seed(0)
    t = torch.randn(2, 1, 1000, generator=g, dtype=torch.float64)
    assert float(io_ref.si_sdr(3.0 * t, t).min()) > 100.0          # scale invariance, p = a t -> "infinite" (eps-limited)
    n = torch.randn(2, 1, 1000, generator=g, dtype=torch.float64)
    n = n - (n * t).sum(-1, keepdim=True) / (t * t).sum(-1, keepdim=True) * t   # orthogonal noise
    snr = 10 * torch.log10((t * t).sum(-1) / (n * n).sum(-1))
    assert torch.allclose(io_ref.si_sdr(t + n, t), snr, atol=1e-9)  # orthogonal noise -> plain SNR
    assert torch.allclose(io_ref.si_sdr(-t + n, t), io_ref.si_sdr(t + n, t), atol=1e-9)  # the projection makes it sign invariant
    # uPIT picks the permutation with the smaller mean neg-SI-SDR: swapped estimates are un-swapped
    tt = torch.randn(3, 2, 800, generator=g, dtype=torch.float64)
    est = tt.flip(1) + 0.01 * torch.randn(3, 2, 800, generator=g, dtype=torch.float64)
    loss, per_item, perm = io_ref.pit_neg_si_sdr(est, tt)
    assert perm.tolist() == [[1, 0]] * 3 and float(loss) < -35.0 and per_item.shape == (3,)


def test_clip_adam(backend):
    n = 10007
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    p = p0.clone().to(backend.device)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    scratch = torch.zeros(300, device=backend.device)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (3.0 if step == 2 else 0.01)  # step 2 is clipped, the others are not
        ref_p.grad = grad.clone()
        norm = torch.nn.utils.clip_grad_norm_([ref_p], 5.0)
        opt.step()
        gdev = grad.clone().to(backend.device)
        ops.clip_adam_step(backend.lib, p, gdev, m, v, scratch, step, lr=1e-3, max_norm=5.0)
        assert abs(float(scratch[0]) - float(norm)) < 1e-4 * float(norm)
        assert float(gdev.abs().max()) == 0.0  # gradient buffer re-zeroed
        assert rel_l2(p, ref_p.detach()) < 1e-6


@pytest.mark.parametrize("decoupled", [False, True])
def test_clip_adam_weight_decay(backend, decoupled):
    """weight decay: torch.optim.Adam adds wd * p to the gradient, torch.optim.AdamW shrinks p by lr * wd (configs of the online
    models use AdamW): the kernel's flag bit 1 selects the decoupled form"""
    n = 4099
    g = torch.Generator().manual_seed(7)
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = (torch.optim.AdamW if decoupled else torch.optim.Adam)([ref_p], lr=1e-2, weight_decay=0.05)
    p = p0.clone().to(backend.device)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    scratch = torch.zeros(300, device=backend.device)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * 0.01
        ref_p.grad = grad.clone()
        opt.step()
        ops.clip_adam_step(backend.lib, p, grad.clone().to(backend.device), m, v, scratch, step, lr=1e-2, weight_decay=0.05, max_norm=0.0,
                           decoupled_weight_decay=decoupled)
        assert rel_l2(p, ref_p.detach()) < 1e-6
