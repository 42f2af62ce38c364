"""Per-phase share of wave time inside one backward kernel (diagnostic build: `python -m nbss_amd.build phase`).
usage: NBSS_HIP_FLAVOUR=phase python tools/phase_prof.py tconvffn_bwd"""
import ctypes as C
import os
import sys
from pathlib import Path

os.environ.setdefault("NBSS_HIP_FLAVOUR", "phase")
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from nbss_amd import ops  # noqa: E402
from nbss_amd._lib import NBSS_BF16, hip, make_cfg  # noqa: E402


def main():
    name = sys.argv[1]
    B, iters = (int(sys.argv[2]) if len(sys.argv) > 2 else 8), 3
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 251  # (kernels whose LDS is full at T = 251 need a shorter T for the timer slots)
    dev = torch.device("cuda:0")
    lib = hip()
    cfg = make_cfg(B, 129, T, 12, 4, L=1, dtype=NBSS_BF16)
    flat = ops.random_params(lib, cfg, dev)
    packed = ops.pack_params(lib, cfg, flat)
    x = torch.randn(B, 129, T, 96, device=dev).bfloat16()
    dy = torch.randn(B, 129, T, 96, device=dev).bfloat16()
    G = torch.zeros_like(flat)
    ws = ops.workspace(lib, cfg, dev)
    o = ops.mhsa_save(lib, cfg, dev)
    tsv = ops.tconvffn_save(lib, cfg, dev)  # the training-mode forward's saved pre-activations (what the product backward reads)
    if tsv is not None:
        ops.tconvffn_fwd(lib, cfg, flat, packed, 0, x, t_save=tsv)
    fns = {
        "fconv_fwd": lambda: ops.fconv_fwd(lib, cfg, flat, packed, 0, 0, x),
        "mhsa_fwd": lambda: ops.mhsa_fwd(lib, cfg, flat, packed, 0, x, o_save=o),
        "tconvffn_fwd": lambda: ops.tconvffn_fwd(lib, cfg, flat, packed, 0, x, t_save=tsv),  # training mode (what the step runs)
        "tconvffn_fwd_infer": lambda: ops.tconvffn_fwd(lib, cfg, flat, packed, 0, x),
        "fconv_bwd": lambda: ops.fconv_bwd(lib, cfg, flat, G, packed, 0, 0, x, dy, ws),
        "full_bwd": lambda: ops.full_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws),
        "mhsa_bwd": lambda: ops.mhsa_bwd(lib, cfg, flat, G, packed, 0, x, dy, o, ws),
        "tconvffn_bwd": lambda: ops.tconvffn_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws, t_save=tsv),
        "tconvffn_bwd_recompute": lambda: ops.tconvffn_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws),
    }
    if name == "mhsa_bwd":
        fns["mhsa_fwd"]()
    reader = getattr(lib, "nbss_phase_read_" + (sys.argv[4] if len(sys.argv) > 4 else name))  # (the single-pass bf16 attention backward shares mhsa_bwd's accumulators)
    reader.restype = C.c_int
    buf = (C.c_ulonglong * 32)()
    fns[name]()
    torch.cuda.synchronize()
    reader(buf)  # discard the warm-up launch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fns[name]()
    e1.record()
    torch.cuda.synchronize()
    n = reader(buf)
    tot = sum(buf[i] for i in range(n)) or 1
    print(f"{name}: {e0.elapsed_time(e1) / iters * 1e3:.0f} us per call incl. wgrad; phase shares of wave time (clock64 ticks):")
    for i in range(n):
        if buf[i]:
            print(f"  phase {i:2d}: {100.0 * buf[i] / tot:5.1f} %   {buf[i] / iters / 1e6:9.2f} Mticks/call")


if __name__ == "__main__":
    main()
