"""Native NB-BLSTM (nbss_amd/blstm.py over nbss_nb_blstm_fwd / _bwd + the dense building blocks) against the torch.nn module it reads its parameters from
(models/arch/blstm2_fc1.py, pinned to the reference's BLSTM2_FC1 by tests/test_nb_models.py): forward, and every parameter gradient against torch.autograd in
fp64."""
import copy

import pytest
import torch

from nbss_amd._lib import NBSS_BF16, NBSS_F32
from util import rel_l2


def _net(hidden=(256, 128), din=4, dout=4):
    from models.arch.blstm2_fc1 import BLSTM2_FC1
    torch.manual_seed(7)
    return BLSTM2_FC1(dim_input=din, dim_output=dout, hidden_size=hidden)


@pytest.mark.parametrize("dtype", [pytest.param(NBSS_F32, id="f32"), pytest.param(NBSS_BF16, id="bf16")])
def test_native_blstm_forward_and_gradients(backend, dtype):
    from nbss_amd.blstm import NativeBLSTM, supported
    hip = backend.name == "hip"
    B, F, T = (2, 129, 63) if hip else (1, 3, 6)  # (BASELINE config 1: 129 frequencies x 63 frames)
    net = _net()
    assert supported(net) is None
    g = torch.Generator().manual_seed(3)
    td = torch.bfloat16 if dtype == NBSS_BF16 else torch.float32
    x = torch.randn(B, F, T, 4, generator=g).to(td)
    r = torch.randn(B, F, T, 4, generator=g)
    ref = copy.deepcopy(net).double()
    want = ref(x.double())
    (want * r.double()).sum().backward()
    net = net.to(backend.device)
    run = NativeBLSTM(net, backend.lib)
    y0 = run.forward(x.to(backend.device))
    tol = 2e-5 if dtype == NBSS_F32 else 3e-2
    assert y0.shape == want.shape and rel_l2(y0, want.detach()) < tol
    y = run.forward_train(x.to(backend.device))
    assert torch.equal(y.detach(), y0)
    (y.float() * r.to(backend.device)).sum().backward()
    gtol = 1e-4 if dtype == NBSS_F32 else 6e-2
    bad = {}
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        e = rel_l2(p.grad, q.grad)
        if e > gtol:
            bad[n] = e
    assert not bad, bad


def test_supported_names_the_reason():
    from nbss_amd.blstm import supported
    assert "hidden size" in supported(_net(hidden=(8, 6)))
    assert supported(_net(hidden=(128, 128))) is None


@pytest.mark.gpu
def test_module_dispatch_on_the_device(hip_lib, monkeypatch):
    """models.arch.blstm2_fc1.BLSTM2_FC1.forward on a HIP tensor: the shipped hidden sizes take the native path (inference and training, silently) and equal
    torch's LSTM on the same device (NBSS_BLSTM_NATIVE=0); other sizes run torch.nn with one warning"""
    import warnings
    net = _net().cuda()
    x = torch.randn(2, 129, 63, 4, generator=torch.Generator().manual_seed(1)).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        y = net(x)
        assert type(y.grad_fn).__name__ == "_BLSTMTrainFnBackward"
        y.square().mean().backward()
        g = {n: p.grad.clone() for n, p in net.named_parameters()}
        with torch.no_grad():
            y_inf = net(x)
    assert torch.equal(y_inf, y.detach())
    monkeypatch.setenv("NBSS_BLSTM_NATIVE", "0")
    net.zero_grad()
    with pytest.warns(RuntimeWarning, match="NBSS_BLSTM_NATIVE=0"):
        yt = net(x)
    yt.square().mean().backward()
    assert rel_l2(y, yt.detach()) < 2e-5
    for n, p in net.named_parameters():
        assert rel_l2(g[n], p.grad) < 2e-4, n
    monkeypatch.delenv("NBSS_BLSTM_NATIVE")
    small = _net(hidden=(16, 8)).cuda()
    with pytest.warns(RuntimeWarning, match="hidden size"):
        assert small(x).shape == (2, 129, 63, 4)
