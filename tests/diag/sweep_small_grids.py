"""Exhaustive walk of the small-grid space the seeded hypothesis tests sample (tests/test_kernels_bwd.py:
test_block_forward_backward_random_small_grids): every fused block x both stream types x B in {1, 2} x F in {1, 2, 3, 5, 17} x T in
{1, 2, 3, 5, 16, 17, 33}, forward + backward against the fp64 oracle, on the emulator build.  ~560 cases; run by hand after touching a fold,
a workspace layout or a tile loop:   python tests/diag/sweep_small_grids.py [nworkers]
"""
import itertools
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]


def run(case):
    import torch
    torch.set_num_threads(1)
    from nbss_amd import ops
    from nbss_amd._lib import NBSS_BF16, NBSS_F32, Lib
    from nbss_amd.build import EMU_LIB
    from oracle import spatialnet_ref as ref
    from conftest import Backend
    from test_kernels_bwd import FULL_NAMES, TF_NAMES, check_param_grads, oracle_grads
    from util import Case, rel_l2
    B, F, T, block, dtype = case
    be = Backend("emu", Lib(EMU_LIB), torch.device("cpu"))
    mh = ["layers.0.norm_mhsa.weight", "layers.0.norm_mhsa.bias", "layers.0.mhsa.in_proj_weight", "layers.0.mhsa.in_proj_bias", "layers.0.mhsa.out_proj.weight",
          "layers.0.mhsa.out_proj.bias"]
    fc = [f"layers.0.fconv1.{k}" for k in ("0.weight", "0.bias", "1.weight", "1.bias", "2.weight")]
    fwd_ref, names, btol = {"fconv": (lambda x, p: ref.fconv(x, p, "layers.0.fconv1"), fc, 5e-2), "full": (lambda x, p: ref.full(x, p, "layers.0"), FULL_NAMES, 3e-2),
                            "mhsa": (lambda x, p: ref.mhsa(x, p, "layers.0"), mh, 3e-2), "tconvffn": (lambda x, p: ref.tconvffn(x, p, "layers.0"), TF_NAMES, 3e-2)}[block]
    try:
        cs = Case(be, B, F, T, dtype)
        x, x64 = cs.stream(seed=11)
        dy, dy64 = cs.stream(seed=111, scale=0.5)
        G = torch.zeros_like(cs.flat)
        ws = ops.workspace(cs.lib, cs.cfg, be.device)
        if block == "mhsa":
            save = ops.mhsa_save(cs.lib, cs.cfg, be.device)
            y = ops.mhsa_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x, o_save=save)
            dx = ops.mhsa_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, save, ws)
        elif block == "fconv":
            y = ops.fconv_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, 0, x)
            dx = ops.fconv_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, 0, x, dy, ws)
        elif block == "full":
            y = ops.full_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
            dx = ops.full_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws)
        else:
            y = ops.tconvffn_fwd(cs.lib, cs.cfg, cs.flat, cs.packed, 0, x)
            dx = ops.tconvffn_bwd(cs.lib, cs.cfg, cs.flat, G, cs.packed, 0, x, dy, ws)
        tol = 1e-4 if dtype == NBSS_F32 else btol
        assert rel_l2(y, fwd_ref(x64, cs.p64)) < (2e-5 if dtype == NBSS_F32 else 1.5e-2), "fwd"
        want_dx, want_g = oracle_grads(fwd_ref, x64, cs.p64, dy64, names)
        assert rel_l2(dx, want_dx) < tol, ("dx", rel_l2(dx, want_dx))
        check_param_grads(cs, G, want_g, tol if T * F * B >= 128 or dtype == NBSS_F32 else 4 * tol)  # (sweep_small_grids: (2,3,16) fconv bf16 lands at 5.1e-2)
        return case, None
    except Exception as e:  # noqa: BLE001
        return case, repr(e)[:300]


if __name__ == "__main__":
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    cases = list(itertools.product([1, 2], [1, 2, 3, 5, 17], [1, 2, 3, 5, 16, 17, 33], ["fconv", "full", "mhsa", "tconvffn"], [0, 1]))
    bad = 0
    with ProcessPoolExecutor(nw) as ex:
        for case, err in ex.map(run, cases, chunksize=4):
            if err:
                bad += 1
                print("FAIL", case, err, flush=True)
    print(f"{len(cases) - bad} / {len(cases)} passed")
    sys.exit(1 if bad else 0)
