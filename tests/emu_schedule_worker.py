"""worker of test_emu_schedules.py: every sub-block kernel once on the host emulator under the fiber schedule named by
HIPEMU_ORDER (read once per process by the emulator), outputs saved for a bit-exact comparison between schedules.
usage: HIPEMU_ORDER=<mode> python tests/emu_schedule_worker.py <out.pt>"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from nbss_amd import ops  # noqa: E402
from nbss_amd._lib import NBSS_BF16, NBSS_F32, Lib  # noqa: E402
from nbss_amd.build import build_emu  # noqa: E402
from util import Case  # noqa: E402


class _BE:
    name, device = "emu", torch.device("cpu")


def main():
    be = _BE()
    be.lib = Lib(build_emu())
    out = {}
    # (1,1,251): all 16 strips / both sequence halves of the narrow-band kernels; (2,33,19): ragged tiles of the cross-band kernels
    for dtype, nm in ((NBSS_BF16, "bf16"), (NBSS_F32, "f32")):
        for (B, F, T) in ((1, 1, 251), (2, 33, 19)):
            cs = Case(be, B, F, T, dtype)
            x, _ = cs.stream(seed=30)
            dy, _ = cs.stream(seed=31)
            key = f"{nm}_{B}x{F}x{T}_"
            save = ops.mhsa_save(cs.lib, cs.cfg, x.device)
            out[key + "fconv_fwd"] = ops.fconv_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, 0, x)
            out[key + "full_fwd"] = ops.full_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
            out[key + "mhsa_fwd"] = ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=save)
            out[key + "tconvffn_fwd"] = ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
            for name, fn in (("fconv_bwd", lambda G, ws: ops.fconv_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, 0, x, dy, ws)),
                             ("full_bwd", lambda G, ws: ops.full_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws)),
                             ("mhsa_bwd", lambda G, ws: ops.mhsa_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, save, ws)),
                             ("tconvffn_bwd", lambda G, ws: ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws))):
                G = torch.zeros_like(cs.flat)
                ws = ops.workspace(cs.lib, cs.cfg, x.device)
                ws.zero_()
                out[key + name] = fn(G, ws)
                out[key + name + "_G"] = G
            tsv = ops.tconvffn_save(cs.lib, cs.cfg, x.device)
            if tsv is not None:  # bf16 stream: training-mode forward + the backward kernel that reads what it saved
                out[key + "tconvffn_fwd_save"] = ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, t_save=tsv)
                G = torch.zeros_like(cs.flat)
                ws = ops.workspace(cs.lib, cs.cfg, x.device)
                ws.zero_()
                out[key + "tconvffn_bwd_saved"] = ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws, t_save=tsv)
                out[key + "tconvffn_bwd_saved_G"] = G
        # long-sequence forward kernels (chunk / key-block boundaries)
        cs = Case(be, 1, 1, 300, dtype)
        x, _ = cs.stream(seed=32)
        out[f"{nm}_long_mhsa"] = ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=ops.mhsa_save(cs.lib, cs.cfg, x.device))
        out[f"{nm}_long_tconvffn"] = ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
        # SpatialNet-large forward kernels (geometry templates; tconvffn_g.hip's chunked two-pass walk, 64/128-key attention blocks)
        cs = Case(be, 1, 3, 40, dtype, geo="large")
        x, _ = cs.stream(seed=33)
        out[f"{nm}_large_fconv"] = ops.fconv_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, 0, x)
        out[f"{nm}_large_full"] = ops.full_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
        out[f"{nm}_large_mhsa"] = ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=ops.mhsa_save(cs.lib, cs.cfg, x.device))
        out[f"{nm}_large_tconvffn"] = ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
        # ... and the geometry-generic backward (gbwd.hip: LDS-staged tap-GEMM weights, LDS partial sums of the row / GroupNorm kernels, the two attention kernels)
        dy, _ = cs.stream(seed=34)
        save = ops.mhsa_save(cs.lib, cs.cfg, x.device)
        for name, fn in (("fconv_bwd", lambda G, ws: ops.fconv_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, 1, x, dy, ws)),
                         ("full_bwd", lambda G, ws: ops.full_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws)),
                         ("mhsa_bwd", lambda G, ws: ops.mhsa_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, save, ws)),
                         ("tconvffn_bwd", lambda G, ws: ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws))):
            G = torch.zeros_like(cs.flat)
            ws = ops.workspace(cs.lib, cs.cfg, x.device)
            ws.zero_()
            out[f"{nm}_large_{name}"] = fn(G, ws)
            out[f"{nm}_large_{name}_G"] = G
    torch.save(out, sys.argv[1])


if __name__ == "__main__":
    main()
