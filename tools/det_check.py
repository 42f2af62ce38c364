"""Run-to-run determinism of every sub-block kernel on the GPU: same inputs twice, outputs compared bit for bit
(weight gradients are compared with a tolerance: their cross-workgroup reductions reorder fp32 sums).
usage: python tools/det_check.py [B] [T]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from nbss_amd import ops  # noqa: E402
from nbss_amd._lib import NBSS_BF16, NBSS_F32, hip, make_cfg  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 251
    dev = torch.device("cuda:0")
    lib = hip()
    for dtype, nm in ((NBSS_BF16, "bf16"), (NBSS_F32, "f32")):
        cfg = make_cfg(B, 129, T, 12, 4, L=1, dtype=dtype)
        flat = ops.random_params(lib, cfg, dev)
        packed = ops.pack_params(lib, cfg, flat)
        sd = torch.bfloat16 if dtype == NBSS_BF16 else torch.float32
        x = torch.randn(B, 129, T, 96, device=dev).to(sd)
        dy = torch.randn(B, 129, T, 96, device=dev).to(sd)
        o = ops.mhsa_save(lib, cfg, dev)
        ws = ops.workspace(lib, cfg, dev)

        def bwd(fn):
            def run():
                G = torch.zeros_like(flat)
                ws.zero_()
                dx = fn(G)
                torch.cuda.synchronize()
                return dx, G
            return run

        ops.mhsa_fwd(lib, cfg, flat, packed, 0, x, o_save=o)
        cases = {
            "fconv_fwd": lambda: (ops.fconv_fwd(lib, cfg, flat, packed, 0, 0, x), None),
            "full_fwd": lambda: (ops.full_fwd(lib, cfg, flat, packed, 0, x), None),
            "mhsa_fwd": lambda: (ops.mhsa_fwd(lib, cfg, flat, packed, 0, x), None),
            "tconvffn_fwd": lambda: (ops.tconvffn_fwd(lib, cfg, flat, packed, 0, x), None),
            "fconv_bwd": bwd(lambda G: ops.fconv_bwd(lib, cfg, flat, G, packed, 0, 0, x, dy, ws)),
            "full_bwd": bwd(lambda G: ops.full_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws)),
            "mhsa_bwd": bwd(lambda G: ops.mhsa_bwd(lib, cfg, flat, G, packed, 0, x, dy, o, ws)),
            "tconvffn_bwd": bwd(lambda G: ops.tconvffn_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws)),
        }
        for name, fn in cases.items():
            worst, wg = 0.0, 0.0
            a, Ga = fn()
            a = a.clone()
            for _ in range(3):
                b, Gb = fn()
                torch.cuda.synchronize()
                worst = max(worst, float((a.float() - b.float()).abs().max()))
                if Ga is not None:
                    wg = max(wg, float((Ga - Gb).norm() / (Ga.norm() + 1e-30)))
            nbad = int((a != b).sum())
            print(f"{nm} {name:13s} out max|diff| {worst:.3e} ({nbad} elements differ in the last pair)  grads rel diff {wg:.2e}", flush=True)


if __name__ == "__main__":
    main()
