"""Time the wgrad launches of one sub-block's backward (HIP-event profiler) — run under NBSS_WG_DEBUG=<bits> to knock out
parts of wgrad_tr3_kernel (diagnostic build: `python -m nbss_amd.build phase`; results are then wrong; this is a where-does-the-time-go probe only)."""
import ctypes as C
import os
import sys
from pathlib import Path

os.environ.setdefault("NBSS_HIP_FLAVOUR", "phase")  # the probe bits only exist in the diagnostic build
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from nbss_amd import ops  # noqa: E402
from nbss_amd._lib import NBSS_BF16, hip, make_cfg  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "tconvffn"
    B, iters = 8, 3
    dev = torch.device("cuda:0")
    lib = hip()
    cfg = make_cfg(B, 129, 251, 12, 4, L=1, dtype=NBSS_BF16)
    flat = ops.random_params(lib, cfg, dev)
    packed = ops.pack_params(lib, cfg, flat)
    x = torch.randn(B, 129, 251, 96, device=dev).bfloat16()
    dy = torch.randn(B, 129, 251, 96, device=dev).bfloat16()
    G = torch.zeros_like(flat)
    ws = ops.workspace(lib, cfg, dev)
    o = ops.mhsa_save(lib, cfg, dev)
    fns = {
        "fconv": lambda: ops.fconv_bwd(lib, cfg, flat, G, packed, 0, 0, x, dy, ws),
        "full": lambda: ops.full_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws),
        "mhsa": lambda: ops.mhsa_bwd(lib, cfg, flat, G, packed, 0, x, dy, o, ws),
        "tconvffn": lambda: ops.tconvffn_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws),
    }
    if name == "mhsa":
        ops.mhsa_fwd(lib, cfg, flat, packed, 0, x, o_save=o)
    fns[name]()
    torch.cuda.synchronize()
    nk = lib.nbss_profile_kernels()
    lib.nbss_profile_enable((1 << nk) - 1)
    for _ in range(iters):
        fns[name]()
    torch.cuda.synchronize()
    ms, cnt = (C.c_double * nk)(), (C.c_int64 * nk)()
    lib.nbss_profile_read(ms, cnt)
    out = {lib.nbss_profile_name(i).decode(): round(ms[i] / iters * 1e3, 1) for i in range(nk) if cnt[i]}
    print(f"NBSS_WG_DEBUG={os.environ.get('NBSS_WG_DEBUG', '0'):>2s} {name}_bwd us/call:", out)


if __name__ == "__main__":
    main()
