"""The C ABI: every symbol include/nbss_hip.h declares is exported by the built libraries and bound by nbss_amd._lib
(no compute calls here: loading libnbss_hip.so needs the HIP runtime but not a GPU)."""
import re
import subprocess
from pathlib import Path

import pytest

from nbss_amd._lib import SIGNATURES, hip_lib_path

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    txt = (ROOT / "include" / "nbss_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nbss_[a-z0-9_]+)\s*\(", txt)))


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_header_matches_python_binding():
    decl = declared_symbols()
    assert decl, "no declarations parsed"
    assert sorted(SIGNATURES) == decl, (sorted(set(decl) - set(SIGNATURES)), sorted(set(SIGNATURES) - set(decl)))


def test_emulator_library_exports_every_symbol(emu_lib):
    syms = exported(emu_lib.path)
    missing = [s for s in declared_symbols() if s not in syms]
    assert not missing, missing
    assert "emulator" in emu_lib.build_info()


def test_gfx950_library_exports_every_symbol():
    from nbss_amd.build import build_hip
    path = build_hip()  # hipcc cross-compiles without a GPU
    syms = exported(path)
    missing = [s for s in declared_symbols() if s not in syms]
    assert not missing, missing
    assert path == hip_lib_path()


def test_product_path_fails_loudly_without_a_gpu(emu_lib):
    """no CPU fallback: the hot path refuses CPU tensors when handed the gfx950 library, and the drop-in module refuses CPU input"""
    import torch
    from models.arch.SpatialNet import SpatialNet
    net = SpatialNet(dim_input=12, dim_output=4, num_layers=1, dim_hidden=96, dim_ffn=192, num_heads=4, dim_squeeze=8, num_freqs=129)
    with pytest.raises(RuntimeError, match="HIP"):
        net(torch.zeros(1, 129, 8, 12))
