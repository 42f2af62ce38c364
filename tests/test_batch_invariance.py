"""Batch invariance of the training walk at the batch the bench times: grid-dependent choices inside the library
(slab widths of the full-band block, tail launches of the forward walk, the gradient stream's priority, one- vs multi-round grids) must not
change what an utterance computes.  On the GPU, with the second stream on:
  * every sub-block kernel, forward and backward: the first two utterances of a batch-32 launch are BITWISE the batch-2 launch's;
  * the walk: utterances 0-1 of the batch-32 training-mode forward are bitwise the batch-2 forward; the batch-32 flat gradient equals the sum
    of the sixteen batch-2 gradients up to the rounding of the fp32 folds / atomics."""
import numpy as np
import pytest
import torch

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, make_cfg
from nbss_amd.engine import SpatialNetEngine
from oracle import spatialnet_ref as ref

F, T = 129, 251


def _blocks(lib, dev, B, x, dy):
    """dict name -> tensor of every sub-block's forward output and backward dx at batch B (inputs: the first B utterances of x / dy)"""
    cfg = make_cfg(B, F, T, 12, 4, L=1, dtype=NBSS_BF16)
    flat = ops.flatten_params(lib, cfg, ref.init_params(num_layers=1, num_freqs=F, seed=1), dev)
    packed = ops.pack_params(lib, cfg, flat)
    xb, dyb = x[:B].contiguous(), dy[:B].contiguous()
    ws = ops.workspace(lib, cfg, dev)
    G = torch.zeros_like(flat)
    out = {}
    out["fconv_fwd"] = ops.fconv_fwd(lib, cfg, flat, packed, 0, 0, xb)
    out["full_fwd"] = ops.full_fwd(lib, cfg, flat, packed, 0, xb)
    o = ops.mhsa_save(lib, cfg, dev)
    out["mhsa_fwd"] = ops.mhsa_fwd(lib, cfg, flat, packed, 0, xb, o_save=o)
    sv = ops.tconvffn_save(lib, cfg, dev)
    out["tconvffn_fwd"] = ops.tconvffn_fwd(lib, cfg, flat, packed, 0, xb, t_save=sv)
    out["fconv_bwd"] = ops.fconv_bwd(lib, cfg, flat, G, packed, 0, 0, xb, dyb, ws)
    out["full_bwd"] = ops.full_bwd(lib, cfg, flat, G, packed, 0, xb, dyb, ws)
    out["mhsa_bwd"] = ops.mhsa_bwd(lib, cfg, flat, G, packed, 0, xb, dyb, o, ws)
    out["tconvffn_bwd"] = ops.tconvffn_bwd(lib, cfg, flat, G, packed, 0, xb, dyb, ws, t_save=sv)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in out.items()}


@pytest.mark.gpu
def test_sub_block_kernels_are_batch_invariant(hip_lib):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    x = torch.randn(32, F, T, 96, generator=g).to(dev).to(torch.bfloat16)
    dy = (0.5 * torch.randn(32, F, T, 96, generator=g)).to(dev).to(torch.bfloat16)
    small = _blocks(hip_lib, dev, 2, x, dy)
    big = _blocks(hip_lib, dev, 32, x, dy)
    bad = [k for k in small if not torch.equal(big[k][:2], small[k])]
    assert not bad, bad


@pytest.mark.gpu
def test_walk_is_batch_invariant_and_gradients_add_up(hip_lib):
    dev = torch.device("cuda:0")
    L = 2
    eng = SpatialNetEngine(hip_lib, dev, dim_input=12, dim_output=4, num_freqs=F, num_layers=L, dtype=NBSS_BF16)
    eng.load_params(ref.init_params(num_layers=L, num_freqs=F, dim_input=12, dim_output=4, seed=0))
    g = torch.Generator().manual_seed(3)
    xin = torch.randn(32, F, T, 12, generator=g).to(dev).to(torch.bfloat16)
    dout = torch.randn(32, F, T, 4, generator=g).to(dev)
    eng.grads.zero_()
    y32 = eng.forward(xin, train=True).clone()
    eng.backward(xin, dout)
    torch.cuda.synchronize()
    g32 = eng.grads.double().cpu().numpy()
    gsum = np.zeros_like(g32)
    for i in range(16):
        eng.grads.zero_()
        y2 = eng.forward(xin[2 * i:2 * i + 2].contiguous(), train=True)
        assert torch.equal(y2, y32[2 * i:2 * i + 2]), i  # bitwise: no reduction crosses utterances in the forward walk
        eng.backward(xin[2 * i:2 * i + 2].contiguous(), dout[2 * i:2 * i + 2].contiguous())
        torch.cuda.synchronize()
        gsum += eng.grads.double().cpu().numpy()
    scale = np.linalg.norm(gsum)
    assert scale > 0 and np.isfinite(g32).all()
    assert np.linalg.norm(g32 - gsum) / scale <= 5e-5, np.linalg.norm(g32 - gsum) / scale
    # per tensor, too: a dropped or doubled partial row of one parameter is invisible in the global norm
    views32, viewss = eng.param_views(torch.from_numpy(g32)), eng.param_views(torch.from_numpy(gsum))
    bad = {}
    for k, v in viewss.items():
        n = float(v.norm())
        if n > 0 and float((views32[k] - v).norm()) / n > 2e-3:  # (the T-conv weight gradients fold bf16 per-sequence partial rows: same rows, another order)
            bad[k] = float((views32[k] - v).norm()) / n
    assert not bad, bad
