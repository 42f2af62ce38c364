"""helpers shared by the kernel parity tests"""
import torch

from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, NBSS_F32, make_cfg
from oracle import spatialnet_ref as ref


def _t(a):
    import numpy as np
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(a)
    a = a.detach().cpu()
    if a.is_complex():
        a = torch.view_as_real(a.to(torch.complex128))
    return a.double()


def rel_l2(a, b):
    a = _t(a)
    b = _t(b)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


class Case:
    """one (cfg, params, packed) bundle on a backend"""

    def __init__(self, backend, B, F, T, dtype, L=1, C_in=12, C_out=4, seed=0):
        self.be = backend
        self.lib = backend.lib
        self.cfg = make_cfg(B, F, T, C_in, C_out, L=L, dtype=dtype)
        self.p = ref.init_params(num_layers=L, num_freqs=F, dim_input=C_in, dim_output=C_out, seed=seed)
        self.p64 = {k: v.double() for k, v in self.p.items()}
        self.flat = ops.flatten_params(self.lib, self.cfg, self.p, backend.device)
        self.packed = ops.pack_params(self.lib, self.cfg, self.flat)
        self.sdtype = ops.stream_dtype(self.cfg)
        # tolerances: fp32 path vs fp64 oracle ; bf16 path vs fp64 oracle fed with bf16-rounded input
        self.tol = 2e-5 if dtype == NBSS_F32 else 1.5e-2

    def stream(self, seed=1, H=96, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        x = scale * torch.randn(self.cfg.B, self.cfg.F, self.cfg.T, H, generator=g)
        xs = x.to(self.sdtype)
        return xs.to(self.be.device), xs.double()
