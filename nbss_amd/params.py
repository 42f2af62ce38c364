"""Names and shapes of SpatialNet's parameters in the order of the flat fp32 buffer
(`nbss_param_table`, include/nbss_hip.h).  The names are the reference's state_dict keys
(models/arch/SpatialNet.py:36-73,175,200) so checkpoints interchange.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

from ._lib import Cfg, Lib

# per-layer suffixes in LayerParam enum order (nbss_amd/csrc/layout.h)
_LAYER = [
    ("fconv1.0.weight", "H"), ("fconv1.0.bias", "H"), ("fconv1.1.weight", "FCW"), ("fconv1.1.bias", "H"), ("fconv1.2.weight", "H"),
    ("norm_full.weight", "H"), ("norm_full.bias", "H"), ("squeeze.0.weight", "SQW"), ("squeeze.0.bias", "SQ"),
    ("full.weight", "FULLW"), ("full.bias", "FULLB"), ("unsqueeze.0.weight", "USQW"), ("unsqueeze.0.bias", "H"),
    ("fconv2.0.weight", "H"), ("fconv2.0.bias", "H"), ("fconv2.1.weight", "FCW"), ("fconv2.1.bias", "H"), ("fconv2.2.weight", "H"),
    ("norm_mhsa.weight", "H"), ("norm_mhsa.bias", "H"), ("mhsa.in_proj_weight", "INW"), ("mhsa.in_proj_bias", "INB"),
    ("mhsa.out_proj.weight", "OUTW"), ("mhsa.out_proj.bias", "H"),
    ("tconvffn.0.weight", "H"), ("tconvffn.0.bias", "H"), ("tconvffn.1.weight", "W1"), ("tconvffn.1.bias", "FFN"),
    ("tconvffn.3.weight", "TCW"), ("tconvffn.3.bias", "FFN"), ("tconvffn.5.weight", "TCW"), ("tconvffn.5.bias", "FFN"),
    ("tconvffn.6.weight", "FFN"), ("tconvffn.6.bias", "FFN"), ("tconvffn.8.weight", "TCW"), ("tconvffn.8.bias", "FFN"),
    ("tconvffn.10.weight", "W2"), ("tconvffn.10.bias", "H"),
]


def _shape(cfg: Cfg, kind: str) -> Tuple[int, ...]:
    H, FFN, SQ, F = cfg.H, cfg.FFN, cfg.SQ, cfg.F
    return {
        "H": (H,), "FFN": (FFN,), "SQ": (SQ,),
        "FCW": (H, H // cfg.f_groups, cfg.f_ks), "SQW": (SQ, H, 1), "USQW": (H, SQ, 1),
        "FULLW": (SQ, F, F), "FULLB": (SQ, F), "INW": (3 * H, H), "INB": (3 * H,), "OUTW": (H, H),
        "W1": (FFN, H, 1), "W2": (H, FFN, 1), "TCW": (FFN, FFN // cfg.t_groups, cfg.t_ks),
    }[kind]


def param_specs(cfg: Cfg) -> List[Tuple[str, Tuple[int, ...]]]:
    """[(state_dict key, shape)] in flat-buffer table order (shared `full` entries repeat)."""
    out = [("encoder.weight", (cfg.H, cfg.C_in, cfg.enc_ks)), ("encoder.bias", (cfg.H,))]
    for l in range(cfg.L):
        for suffix, kind in _LAYER:
            out.append((f"layers.{l}.{suffix}", _shape(cfg, kind)))
    out += [("decoder.weight", (cfg.C_out, cfg.H)), ("decoder.bias", (cfg.C_out,))]
    return out


def param_table(lib: Lib, cfg: Cfg) -> Dict[str, Tuple[int, Tuple[int, ...]]]:
    """{key: (offset in floats, shape)} as reported by the C library, cross-checked against param_specs."""
    specs = param_specs(cfg)
    n = lib.nbss_param_table(C.byref(cfg), None, None, 0)
    if n != len(specs):
        raise RuntimeError(f"nbss_param_table reports {n} entries, python expects {len(specs)}")
    offs = (C.c_int64 * n)()
    nums = (C.c_int64 * n)()
    rc = lib.nbss_param_table(C.byref(cfg), offs, nums, n)
    if rc != n:
        raise RuntimeError(f"nbss_param_table failed: {rc}")
    table = {}
    for i, (name, shape) in enumerate(specs):
        numel = 1
        for s in shape:
            numel *= s
        if numel != nums[i]:
            raise RuntimeError(f"{name}: numel mismatch {numel} vs {nums[i]}")
        table[name] = (int(offs[i]), shape)
    return table
