"""NBC2 inference (BASELINE config 4's network: 8 layers, 96 / 192, 2 heads, 8 ch -> 3 spk, fp32) on the device: native forward (nbss_amd/nbc2.py over the
nbss_nb_* building blocks) against the torch.nn modules (ATen / MIOpen kernels).  usage: python tools/nbc2_throughput.py [batch] [reps]"""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from models.arch.NBC2 import NBC2  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = NBC2(dim_input=16, dim_output=6, n_layers=8, dim_hidden=96, dim_ffn=192, num_freqs=129).to(dev).eval()
    x = torch.randn(B, 129, 251, 16, device=dev)
    out = {}
    with torch.no_grad():
        y_native = net(x)
        with torch.enable_grad():  # (grad mode on: the module takes its torch.nn path; nothing is differentiated)
            y_torch = net(x).detach()
        out["rel_l2_native_vs_torch"] = float((y_native - y_torch).norm() / y_torch.norm())
        for name, ctx in (("native", torch.no_grad()), ("torch_nn", torch.enable_grad())):
            with ctx:
                net(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    y = net(x)
                    y = y.detach()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
            out[name] = {"ms_per_batch": round(dt * 1e3, 2), "utt_per_s": round(B / dt, 1)}
    print(json.dumps({"what": "NBC2 forward, 8 layers, 129 x 251, 8 ch -> 3 spk, fp32", "batch": B, **out}))


if __name__ == "__main__":
    main()
