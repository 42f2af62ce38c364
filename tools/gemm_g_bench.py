"""Dense linear maps of SpatialNet-large at batch 4 (129 516 tokens) through nbss_nb_conv_t (bf16, one tap): microseconds and TFLOP/s per problem.
NBSS_GEMM_V1=1 selects the previous kernel (gb_tap_gemm_lds_kernel) for comparison.  usage: python tools/gemm_g_bench.py [reps]"""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from nbss_amd import ops  # noqa: E402
from nbss_amd._lib import NBSS_BF16, hip  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    lib, dev = hip(), torch.device("cuda:0")
    nseq, T = 516, 251
    out = {"kernel": "v1" if os.environ.get("NBSS_GEMM_V1") == "1" else "tile", "rows": nseq * T}
    for K, M, act in ((192, 576, 0), (192, 192, 0), (576, 192, 0), (192, 384, 1), (384, 192, 0)):
        x = torch.randn(nseq, T, K, device=dev).to(torch.bfloat16)
        w = torch.randn(M, K, 1, device=dev) / K ** 0.5
        b = torch.randn(M, device=dev)
        y = torch.empty(nseq, T, M, dtype=torch.bfloat16, device=dev)
        ws = torch.empty(lib._dll.nbss_nb_ws_bytes(M, K, 1, 1), dtype=torch.uint8, device=dev)

        def run():
            lib.call("nbss_nb_conv_t", NBSS_BF16, nseq, T, K, K, M, 1, 1, ops._ptr(lib, x), ops._ptr(lib, w), ops._ptr(lib, b), ops._ptr(lib, y), None, 0, act,
                     ops._ptr(lib, ws), ops._stream(lib, x))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out[f"{K}->{M}"] = {"us": round(us, 1), "tflops": round(2 * nseq * T * K * M / us / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
