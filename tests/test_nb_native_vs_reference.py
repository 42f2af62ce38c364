"""The native narrow-band paths (nbss_amd/nbc2.py, nbc.py, blstm.py over the nbss_nb_* building blocks) against numbers produced by the REFERENCE's own
modules (/root/reference/models/arch/{NBC2,NBC,blstm2_fc1}.py run in fp64 by tests/golden/make_golden.py nbnative -> tests/golden/nb_models_native.npz) at
the smallest widths the kernels take: output and every parameter gradient of sum(y * r).  The other native tests compare with this repo's torch.nn modules
(themselves pinned to the reference by tests/test_nb_models.py); this one has no such link in between.  Both backends: the host emulator build of the
kernel sources (-m "not gpu") and libnbss_hip.so on the MI355X (-m gpu) read the same committed fixture."""
from pathlib import Path

import numpy as np
import pytest
import torch

from util import rel_l2

GOLD = Path(__file__).resolve().parent / "golden" / "nb_models_native.npz"


def _case(name):
    d = np.load(GOLD)
    pre = name + "/"
    t = {k[len(pre):]: torch.from_numpy(d[k].astype(np.float32) if d[k].dtype == np.float16 else d[k]) for k in d.files if k.startswith(pre)}
    params = {k[len("param/"):]: v for k, v in t.items() if k.startswith("param/")}
    grads = {k[len("grad/"):]: v for k, v in t.items() if k.startswith("grad/")}
    return t["x"], t["y"], t["r"], params, grads


def _load(net, params, skip=()):
    own = net.state_dict()
    assert {k for k in own if not k.endswith(skip)} == set(params), sorted(set(own) ^ set(params))[:6]
    missing, unexpected = net.load_state_dict(params, strict=False)
    assert not unexpected and all(k.endswith(skip) for k in missing), (missing, unexpected)
    return net


def _check(net, y, y_ref, grads, ytol, gtol, floor, every=None):
    assert y.shape == y_ref.shape and rel_l2(y, y_ref) < ytol, rel_l2(y, y_ref)
    top = max(float(g.norm()) for g in grads.values())
    bad, seen = {}, set()
    for n, p in net.named_parameters():
        assert p.grad is not None and n in grads, n
        g = p.grad.detach().float().cpu()
        if every and g.dim() == 2 and g.shape[0] >= 512:  # (the fixture keeps every eighth row of the big matrices)
            g = g[::every]
        want = grads[n]
        assert g.shape == want.shape, (n, g.shape, want.shape)
        err = float((g.double() - want.double()).norm())
        if err > gtol * float(want.norm()) + floor * top:
            bad[n] = (err, float(want.norm()))
        seen.add(n)
    assert seen == set(grads) and not bad, bad


def test_native_nbc2_equals_the_reference(backend):
    from models.arch.NBC2 import NBC2
    from nbss_amd.nbc2 import NativeNBC2, supported
    emu_lib, dev = backend.lib, backend.device
    x, y_ref, r, params, grads = _case("nbc2")
    x, r = x.to(dev), r.to(dev)
    bk = {"n_heads": 2, "dropout": 0, "conv_kernel_size": 3, "n_conv_groups": 4, "norms": ("LN", "GBN", "GBN"),
          "group_batch_norm_kwargs": {"share_along_sequence_dim": False}}
    net = _load(NBC2(dim_input=4, dim_output=4, n_layers=2, dim_hidden=48, dim_ffn=64, num_freqs=5, block_kwargs=bk), params).float().to(dev).train()
    assert supported(net) is None
    run = NativeNBC2(net, emu_lib)
    assert rel_l2(run.forward(x), y_ref) < 5e-6  # (measured: 2e-7; gradients 5e-7)
    y = run.forward_train(x)
    (y.float() * r).sum().backward()
    _check(net, y.detach(), y_ref, grads, 5e-6, 2e-5, 1e-7)


def test_native_nbc_equals_the_reference(backend):
    from models.arch.NBC import NBC
    from nbss_amd.nbc import NativeNBC, train_supported
    emu_lib, dev = backend.lib, backend.device
    x, y_ref, r, params, grads = _case("nbc")
    x, r = x.to(dev), r.to(dev)
    net = _load(NBC(dim_input=4, dim_output=4, n_layers=2, encoder_kernel_size=4, n_heads=2, hidden_size=48, ffn_size=64), params, skip=("rel_pos.pe",)).float()
    for m in net.modules():  # (the fixture is the reference in eval mode: its dropouts are inactive)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net.to(dev).train()
    assert train_supported(net) is None
    run = NativeNBC(net, emu_lib)
    assert rel_l2(run.forward(x), y_ref) < 5e-6  # (measured: 4e-7; gradients 1.4e-6)
    y = run.forward_train(x)
    (y.float() * r).sum().backward()
    _check(net, y.detach(), y_ref, grads, 5e-6, 2e-5, 1e-7)  # (floor: the key bias is gradient-free under the softmax)


def test_native_blstm_equals_the_reference(backend):
    from models.arch.blstm2_fc1 import BLSTM2_FC1
    from nbss_amd.blstm import NativeBLSTM, supported
    emu_lib, dev = backend.lib, backend.device
    x, y_ref, r, params, grads = _case("blstm")
    x, r = x.to(dev), r.to(dev)
    net = _load(BLSTM2_FC1(dim_input=4, dim_output=4, hidden_size=(128, 128)), params).float().to(dev)
    assert supported(net) is None
    run = NativeBLSTM(net, emu_lib)
    assert rel_l2(run.forward(x), y_ref) < 5e-6  # (measured: 4e-7; gradients 4e-7)
    y = run.forward_train(x)
    (y.float() * r).sum().backward()
    _check(net, y.detach(), y_ref, grads, 5e-6, 2e-5, 1e-7, every=8)


@pytest.mark.parametrize("name,cls", [("nbc2", "NBC2"), ("nbc", "NBC"), ("blstm", "BLSTM2_FC1")])
def test_the_torch_modules_equal_the_reference_at_these_widths(name, cls):
    """the same fixture against this repo's drop-in torch.nn modules (the CPU path of models/arch/*): fp64, so the bar is rounding of the stored fp32 values"""
    import importlib
    x, y_ref, r, params, grads = _case(name)
    mod = importlib.import_module("models.arch." + {"NBC2": "NBC2", "NBC": "NBC", "BLSTM2_FC1": "blstm2_fc1"}[cls])
    if name == "nbc2":
        bk = {"n_heads": 2, "dropout": 0, "conv_kernel_size": 3, "n_conv_groups": 4, "norms": ("LN", "GBN", "GBN"),
              "group_batch_norm_kwargs": {"share_along_sequence_dim": False}}
        net = mod.NBC2(dim_input=4, dim_output=4, n_layers=2, dim_hidden=48, dim_ffn=64, num_freqs=5, block_kwargs=bk)
    elif name == "nbc":
        net = mod.NBC(dim_input=4, dim_output=4, n_layers=2, encoder_kernel_size=4, n_heads=2, hidden_size=48, ffn_size=64)
    else:
        net = mod.BLSTM2_FC1(dim_input=4, dim_output=4, hidden_size=(128, 128))
    net = _load(net, params, skip=("rel_pos.pe",)).double().eval()
    y = net(x.double())
    (y * r.double()).sum().backward()
    _check(net, y.detach(), y_ref, grads, 1e-6, 1e-5, 1e-7, every=8 if name == "blstm" else None)
