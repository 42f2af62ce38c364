"""pytest configuration: `gpu` marker, backends.

Two backends run the SAME kernel sources through the SAME C ABI:
  * "emu" — host-emulator build (tests/hipemu), CPU tensors, no GPU needed;  -m "not gpu"
  * "hip" — libnbss_hip.so on a real MI355X;                                   -m gpu
"""
import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

REFERENCE = Path("/root/reference")


os.environ.setdefault("NBSS_POISON_SCRATCH", "1")  # scratch buffers start as NaN bytes in every test (nbss_amd/ops.py: scratch)


# the randomised shape tests draw the SAME examples in every run (round-5 review: an un-seeded draw hit a single-token defect in one run of three);
# the defect classes they found are pinned as explicit @example cases next to each @given, tests/diag/sweep_small_grids.py walks the whole grid
try:
    from hypothesis import settings as _hyp_settings
    _hyp_settings.register_profile("repro", derandomize=True, database=None, deadline=None, print_blob=True)
    _hyp_settings.load_profile("repro")
except ImportError:  # hypothesis is test-only
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running emulator case")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def emu_lib():
    from nbss_amd.build import build_emu
    from nbss_amd._lib import Lib
    return Lib(build_emu())


@pytest.fixture(scope="session")
def hip_lib():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from nbss_amd._lib import hip
    return hip()


class Backend:
    def __init__(self, name, lib, device):
        self.name, self.lib, self.device = name, lib, device


def _backend_params():
    return [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=_backend_params())
def backend(request):
    if request.param == "emu":
        return Backend("emu", request.getfixturevalue("emu_lib"), torch.device("cpu"))
    return Backend("hip", request.getfixturevalue("hip_lib"), torch.device("cuda:0"))


@pytest.fixture(scope="session")
def reference_available():
    return (REFERENCE / "models" / "arch" / "SpatialNet.py").exists()
