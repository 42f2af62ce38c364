"""T-ConvFFN backward: recomputing kernel vs the saved-pre-activation kernel, per-tensor rel-L2 against the fp64 oracle (emulator or HIP).
usage: python tests/diag/diag_tcf_saved.py [emu|hip] B F T"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("NBSS_POISON_SCRATCH", "1")
import torch
from nbss_amd import ops
from nbss_amd._lib import NBSS_BF16, Lib, hip
from nbss_amd.params import param_table
from oracle import spatialnet_ref as ref
from util import Case, rel_l2
from test_kernels_bwd import TF_NAMES, oracle_grads


class BE:
    pass


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "emu"
    B, F, T = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (1, 5, 19)
    be = BE()
    if which == "emu":
        from nbss_amd.build import build_emu
        be.name, be.lib, be.device = "emu", Lib(build_emu()), torch.device("cpu")
    else:
        be.name, be.lib, be.device = "hip", hip(), torch.device("cuda:0")
    cs = Case(be, B, F, T, NBSS_BF16)
    x, x64 = cs.stream(seed=20)
    dy, dy64 = cs.stream(seed=120, scale=0.5)
    want_dx, want_g = oracle_grads(lambda xx, pp: ref.tconvffn(xx, pp, "layers.0"), x64, cs.p64, dy64, TF_NAMES)
    table = param_table(cs.lib, cs.cfg)
    for label in ("recompute", "saved"):
        G = torch.zeros_like(cs.flat)
        ws = ops.workspace(cs.lib, cs.cfg, be.device)
        sv = None
        if label == "saved":
            sv = ops.tconvffn_save(cs.lib, cs.cfg, be.device)
            ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, t_save=sv)
        dx = ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws, t_save=sv)
        print(f"{label:10s} dx {rel_l2(dx, want_dx):.4f}  branch {rel_l2(dx.double().cpu() - dy64, want_dx - dy64):.4f}  " +
              " ".join(f"{n.split('tconvffn.')[1]}={rel_l2(G[table[n][0]:table[n][0] + g.numel()].reshape(table[n][1]), g):.4f}" for n, g in want_g.items()))


main()
