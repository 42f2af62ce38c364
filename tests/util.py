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

    GEO = {"small": dict(H=96, FFN=192, SQ=8), "large": dict(H=192, FFN=384, SQ=16)}  # configs/SpatialNet.yaml:16-24 and its "for large" comments

    def __init__(self, backend, B, F, T, dtype, L=1, C_in=12, C_out=4, seed=0, geo="small"):
        self.be = backend
        self.lib = backend.lib
        g = self.GEO[geo]
        self.cfg = make_cfg(B, F, T, C_in, C_out, L=L, dtype=dtype, **g)
        self.p = ref.init_params(num_layers=L, num_freqs=F, dim_input=C_in, dim_output=C_out, seed=seed, dim_hidden=g["H"], dim_ffn=g["FFN"],
                                 dim_squeeze=g["SQ"])
        self.p64 = {k: v.double() for k, v in self.p.items()}
        self.flat = ops.flatten_params(self.lib, self.cfg, self.p, backend.device)
        self.packed = ops.pack_params(self.lib, self.cfg, self.flat)
        self.sdtype = ops.stream_dtype(self.cfg)
        # tolerances: fp32 path vs fp64 oracle ; bf16 path vs fp64 oracle fed with bf16-rounded input
        self.tol = 2e-5 if dtype == NBSS_F32 else 1.5e-2

    def stream(self, seed=1, H=None, scale=1.0):
        H = self.cfg.H if H is None else H
        g = torch.Generator().manual_seed(seed)
        x = scale * torch.randn(self.cfg.B, self.cfg.F, self.cfg.T, H, generator=g)
        xs = x.to(self.sdtype)
        return xs.to(self.be.device), xs.double()
