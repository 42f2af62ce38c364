"""Run ONE sub-block kernel (fwd or bwd) a few times at the BASELINE geometry — target for rocprofv3 --pmc passes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from nbss_amd import ops  # noqa: E402
from nbss_amd._lib import NBSS_BF16, hip, make_cfg  # noqa: E402


def main():
    name = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
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
    tsv = ops.tconvffn_save(lib, cfg, dev)  # the training-mode forward's saved pre-activations (what the product backward reads)
    if tsv is not None:
        ops.tconvffn_fwd(lib, cfg, flat, packed, 0, x, t_save=tsv)
    fns = {
        "fconv_fwd": lambda: ops.fconv_fwd(lib, cfg, flat, packed, 0, 0, x),
        "full_fwd": lambda: ops.full_fwd(lib, cfg, flat, packed, 0, x),
        "mhsa_fwd": lambda: ops.mhsa_fwd(lib, cfg, flat, packed, 0, x, o_save=o),
        "tconvffn_fwd": lambda: ops.tconvffn_fwd(lib, cfg, flat, packed, 0, x, t_save=tsv),  # training mode (what the step runs)
        "tconvffn_fwd_infer": lambda: ops.tconvffn_fwd(lib, cfg, flat, packed, 0, x),
        "fconv_bwd": lambda: ops.fconv_bwd(lib, cfg, flat, G, packed, 0, 0, x, dy, ws),
        "full_bwd": lambda: ops.full_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws),
        "mhsa_bwd": lambda: ops.mhsa_bwd(lib, cfg, flat, G, packed, 0, x, dy, o, ws),
        "tconvffn_bwd": lambda: ops.tconvffn_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws, t_save=tsv),
        "tconvffn_bwd_recompute": lambda: ops.tconvffn_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws),
    }
    if name.startswith("mhsa_bwd"):
        fns["mhsa_fwd"]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fns[name]()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fns[name]()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name} B={B}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per call (incl. wgrad launches for bwd)")


if __name__ == "__main__":
    main()
