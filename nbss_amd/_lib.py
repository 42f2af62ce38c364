"""ctypes binding of include/nbss_hip.h.

`hip()` returns the gfx950 library (nbss_amd/lib/libnbss_hip.so) and raises loudly when it is
missing or cannot be loaded: there is NO CPU / eager fallback for the hot path.  `Lib(path)` can
also wrap the host-emulator flavour of the same C ABI, which only tests/ may do.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

NBSS_F32, NBSS_BF16 = 0, 1

_ERR = {0: "OK", -1: "NBSS_EINVAL (bad argument)", -2: "NBSS_EUNSUPPORTED (no kernel for this shape/config)",
        -3: "NBSS_ELAUNCH (HIP launch failed)", -4: "NBSS_ELDS (cannot raise dynamic LDS limit)"}


class NbssError(RuntimeError):
    pass


class Cfg(C.Structure):
    """mirror of `nbss_cfg` (include/nbss_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "B", "F", "T", "C_in", "C_out", "H", "FFN", "SQ", "L", "heads", "enc_ks", "f_ks", "t_ks",
        "f_groups", "t_groups", "full_share", "dtype")]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


def make_cfg(B, F, T, C_in, C_out, H=96, FFN=192, SQ=8, L=8, heads=4, enc_ks=5, f_ks=5, t_ks=3,
             f_groups=8, t_groups=8, full_share=0, dtype=NBSS_BF16) -> Cfg:
    return Cfg(B, F, T, C_in, C_out, H, FFN, SQ, L, heads, enc_ks, f_ks, t_ks, f_groups, t_groups, full_share, dtype)


_P = C.c_void_p
_CP = C.POINTER(Cfg)
_I = C.c_int

# name -> (restype, argtypes); every symbol include/nbss_hip.h declares must be listed here
SIGNATURES = {
    "nbss_param_table": (_I, [_CP, _P, _P, _I]),
    "nbss_param_count": (C.c_int64, [_CP]),
    "nbss_packed_bytes": (C.c_int64, [_CP]),
    "nbss_pack_params": (_I, [_CP, _P, _P, _P]),
    "nbss_encoder_fwd": (_I, [_CP, _P, _P, _P, _P, _P]),
    "nbss_decoder_fwd": (_I, [_CP, _P, _P, _P, _P, _P]),
    "nbss_fconv_fwd": (_I, [_CP, _P, _P, _I, _I, _P, _P, _P]),
    "nbss_full_fwd": (_I, [_CP, _P, _P, _I, _P, _P, _P]),
    "nbss_mhsa_fwd": (_I, [_CP, _P, _P, _I, _P, _P, _P, _P]),
    "nbss_tconvffn_save_bytes": (C.c_int64, [_CP]),
    "nbss_tconvffn_fwd": (_I, [_CP, _P, _P, _I, _P, _P, _P, _P]),
    "nbss_workspace_bytes": (C.c_int64, [_CP]),
    "nbss_mhsa_save_bytes": (C.c_int64, [_CP]),
    "nbss_tconvffn_bwd": (_I, [_CP, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "nbss_mhsa_bwd": (_I, [_CP, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "nbss_fconv_bwd": (_I, [_CP, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "nbss_full_bwd": (_I, [_CP, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "nbss_decoder_bwd": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_encoder_bwd": (_I, [_CP, _P, _P, _P, _P]),
    "nbss_acts_bytes": (C.c_int64, [_CP]),
    "nbss_train_ws_bytes": (C.c_int64, [_CP]),
    "nbss_spatialnet_fwd": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_spatialnet_bwd": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_spatialnet_bwd_range": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "nbss_stft_tables_bytes": (C.c_int64, [_I]),
    "nbss_stft_tables": (_I, [_I, _I, _P, _P]),
    "nbss_stft_norm_fwd": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "nbss_istft_ws_bytes": (C.c_int64, [_I, _I, _I, _I]),
    "nbss_inorm_istft_fwd": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "nbss_inorm_istft_bwd": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "nbss_pit_ws_bytes": (C.c_int64, [_I, _I]),
    "nbss_pit_neg_sisdr": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_online_encoder_step": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "nbss_online_ret_step": (_I, [_I, _I] + [_P] * 12),
    "nbss_online_mhsa_step": (_I, [_I, _I, _I, _I] + [_P] * 11),
    "nbss_online_advance": (_I, [_P, _I, _P]),
    "nbss_online_tconvffn_step": (_I, [_I, _I, _I] + [_P] * 21),
    "nbss_nb_ws_bytes": (C.c_int64, [_I, _I, _I, _I]),
    "nbss_nb_conv_t": (_I, [_I, C.c_int64, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "nbss_nb_layernorm": (_I, [_I, C.c_int64, _I, _P, _P, _P, _P, _P, _P]),
    "nbss_nb_group_batch_norm": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, C.c_float, _I, _P, _P]),
    "nbss_nb_attention_fwd": (_I, [_I, C.c_int64, _I, _I, _I, _P, _P, _P]),
    "nbss_nb_attention_relpos_fwd": (_I, [_I, C.c_int64, _I, _I, _I, _P, _P, _P, _P, C.c_float, _P, _P]),
    "nbss_nb_group_norm": (_I, [_I, C.c_int64, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "nbss_nb_bwd_ws_bytes": (C.c_int64, [_I, _I, _I, _I]),
    "nbss_nb_conv_t_train": (_I, [_I, C.c_int64, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_nb_conv_t_bwd": (_I, [_I, C.c_int64, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_nb_layernorm_bwd": (_I, [_I, C.c_int64, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_nb_group_batch_norm_bwd": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, C.c_float, _I, _P, _P, _P, _P, _P]),
    "nbss_nb_attention_bwd_ws_bytes": (C.c_int64, [_I, C.c_int64, _I, _I, _I]),
    "nbss_nb_attention_bwd": (_I, [_I, C.c_int64, _I, _I, _I, _P, _P, _P, _P, _P]),
    "nbss_nb_attention_relpos_train": (_I, [_I, C.c_int64, _I, _I, _I, _P, _P, _P, _P, C.c_float, _P, C.c_float, _P, _P]),
    "nbss_nb_attention_relpos_bwd_ws_bytes": (C.c_int64, [C.c_int64, _I, _I, _I]),
    "nbss_nb_attention_relpos_bwd": (_I, [_I, C.c_int64, _I, _I, _I, _P, _P, _P, _P, C.c_float, _P, C.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_nb_group_norm_train": (_I, [_I, C.c_int64, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
    "nbss_nb_group_norm_bwd": (_I, [_I, C.c_int64, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_nb_blstm_ws_bytes": (C.c_int64, [_I, _I]),
    "nbss_nb_blstm_fwd": (_I, [_I, C.c_int64, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_nb_blstm_bwd": (_I, [_I, C.c_int64, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "nbss_clip_adam_step": (_I, [C.c_int64, _P, _P, _P, _P, _P] + [C.c_float] * 7 + [_I, _I, _P]),
    "nbss_clip_adam_step_dev": (_I, [C.c_int64, _P, _P, _P, _P, _P, _P] + [C.c_float] * 6 + [_I, _P]),
    "nbss_adam_hyper": (_I, [_I, C.c_float, C.c_float, C.c_float, _P]),
    "nbss_selftest_mma": (_I, [_I, _I, _P, _P, _P, _P]),
    "nbss_build_info": (C.c_char_p, []),
    "nbss_profile_enable": (_I, [C.c_int64]),
    "nbss_profile_kernels": (_I, []),
    "nbss_profile_name": (C.c_char_p, [_I]),
    "nbss_profile_read": (_I, [_P, _P]),
}


class Lib:
    def __init__(self, path):
        self.path = str(path)
        self._dll = C.CDLL(self.path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self._dll, name)  # AttributeError when a declared symbol is missing
            fn.restype = res
            fn.argtypes = args

    def __getattr__(self, name):
        if name.startswith("nbss_"):
            return getattr(self._dll, name)
        raise AttributeError(name)

    def call(self, name, *args):
        """call an int-returning entry point and raise on a non-zero code"""
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            raise NbssError(f"{name} failed: {_ERR.get(rc, rc)}")

    def build_info(self) -> str:
        return self._dll.nbss_build_info().decode()


_HIP = None


def hip_lib_path() -> Path:
    # NBSS_HIP_FLAVOUR=phase selects the diagnostic build with in-kernel phase timers (tools/phase_prof.py); any other value
    # names a side-by-side build lib/libnbss_hip_<flavour>.so for A/B timing by the tools (never set by the product)
    flavour = os.environ.get("NBSS_HIP_FLAVOUR", "")
    return Path(__file__).resolve().parent / "lib" / (f"libnbss_hip_{flavour}.so" if flavour else "libnbss_hip.so")


def hip() -> Lib:
    """The product library.  Fails loudly when absent — no fallback path exists."""
    global _HIP
    if _HIP is None:
        p = hip_lib_path()
        if not p.exists():
            raise NbssError(f"{p} is missing: build it with `python -m nbss_amd.build hip` (hipcc, gfx950). "
                            "nbss_amd has no CPU/eager fallback for the SpatialNet hot path.")
        import torch  # noqa: F401  (load torch's HIP runtime first so that we share it)
        _HIP = Lib(p)
    return _HIP
