"""The path a Lightning user hits: TrainModule.forward / training_step (SharedTrainer.py:104-149 of the reference) =
fused STFT+norm -> _SpatialNetFn (one autograd node) -> _FusedIO (inorm + iSTFT) -> models.io.loss.Loss (uPIT neg-SI-SDR),
under autocast for 'bf16-mixed'.  Its loss and parameter gradients must equal what engine.TrainStep (the fused step `fit` and
bench.py run) computes for the same batch."""
import pytest
import torch

from nbss_amd._lib import NBSS_BF16, NBSS_F32


def _module(precision):
    from SharedTrainer import TrainModule
    from models.arch.SpatialNet import SpatialNet
    from models.io.loss import Loss, neg_si_sdr
    from models.io.norm import Norm
    from models.io.stft import STFT
    torch.manual_seed(2)
    arch = SpatialNet(dim_input=12, dim_output=4, num_layers=2, encoder_kernel_size=5, dim_hidden=96, dim_ffn=192, num_heads=4, dropout=(0, 0, 0),
                      kernel_size=(5, 3), conv_groups=(8, 8), norms=("LN", "LN", "GN", "LN", "LN", "LN"), dim_squeeze=8, num_freqs=129, full_share=0)
    m = TrainModule(arch, channels=[0, 1, 2, 3, 4, 5], ref_channel=0, stft=STFT(256, 128, 256), norm=Norm("frequency", online=True),
                    loss=Loss(neg_si_sdr, pit=True)).to("cuda:0")
    m.precision = precision
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["32", "bf16-mixed"])
def test_training_step_matches_fused_step(hip_lib, precision):
    from nbss_amd.engine import TrainStep
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, C, S, N = 2, 6, 2, 8000
    src = torch.randn(B, S, N, generator=g)
    ys = torch.stack([src * (0.6 + 0.1 * c) for c in range(C)], 2)  # [B,S,C,N]
    x = ys.sum(1) + 0.05 * torch.randn(B, C, N, generator=g)
    m = _module(precision)
    loss = m.training_step((x.to(dev), ys.to(dev), None))
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in m.arch.named_parameters()}
    assert all(torch.isfinite(gv).all() for gv in grads.values())
    # the fused step on the same weights and batch
    eng = m.arch._engine_for(dev)
    eng.dtype = NBSS_BF16 if precision == "bf16-mixed" else NBSS_F32
    ts = TrainStep(eng)
    l2, _, dout, xin, _ = ts.forward_loss(x.to(dev), ys[:, :, 0].contiguous().to(dev))
    eng.grads.zero_()
    eng.backward(xin, dout)
    views = eng.param_views(eng.grads)
    assert abs(float(loss) - float(l2)) <= 1e-5 * max(1.0, abs(float(l2)))
    for k, gv in grads.items():
        ref = views[k]
        denom = float(ref.norm()) + 1e-30
        assert float((gv - ref).norm()) / denom <= 1e-4, k  # same kernels; only the weight-gradient atomics reorder sums


@pytest.mark.gpu
def test_two_forwards_before_backward_raise(hip_lib):
    """the engine keeps ONE set of saved activations: a backward that would replay another forward's activations must raise"""
    dev = torch.device("cuda:0")
    m = _module("32")
    g = torch.Generator().manual_seed(6)
    X1 = torch.randn(1, 129, 32, 12, generator=g).to(dev)
    X2 = torch.randn(1, 129, 32, 12, generator=g).to(dev)
    y1 = m.arch(X1)
    y2 = m.arch(X2)
    with pytest.raises(RuntimeError):
        (y1.sum() + y2.sum()).backward()
