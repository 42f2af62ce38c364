"""Time each forward/backward kernel at the BASELINE geometry with HIP events (GPU box only)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from nbss_amd import ops  # noqa: E402
from nbss_amd._lib import NBSS_BF16, NBSS_F32, hip, make_cfg  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    lib = hip()
    out = {}
    for dname, dt in (("bf16", NBSS_BF16), ("f32", NBSS_F32)):
        cfg = make_cfg(B, 129, 251, 12, 4, L=1, dtype=dt)
        flat = ops.random_params(lib, cfg, dev)
        packed = ops.pack_params(lib, cfg, flat)
        sd = ops.stream_dtype(cfg)
        x = torch.randn(B, 129, 251, 96, device=dev).to(sd)
        xin = torch.randn(B, 129, 251, 12, device=dev).to(sd)
        S = x.numel() * x.element_size()
        res = {}
        res["pack"] = timeit(lambda: ops.pack_params(lib, cfg, flat, packed))
        res["encoder"] = timeit(lambda: ops.encoder_fwd(lib, cfg, flat, packed, xin))
        res["fconv"] = timeit(lambda: ops.fconv_fwd(lib, cfg, flat, packed, 0, 0, x))
        res["full"] = timeit(lambda: ops.full_fwd(lib, cfg, flat, packed, 0, x))
        res["mhsa"] = timeit(lambda: ops.mhsa_fwd(lib, cfg, flat, packed, 0, x))
        res["tconvffn"] = timeit(lambda: ops.tconvffn_fwd(lib, cfg, flat, packed, 0, x))
        res["decoder"] = timeit(lambda: ops.decoder_fwd(lib, cfg, flat, packed, x))
        layer = res["fconv"] * 2 + res["full"] + res["mhsa"] + res["tconvffn"]
        fwd_utt = (8 * layer + res["encoder"] + res["decoder"]) / B
        out[dname] = {"B": B, "us": res, "stream_bytes": S, "layer_fwd_us": layer, "fwd_us_per_utt(8 layers)": fwd_utt,
                      "GBps_2S": {k: 2 * S / (v * 1e-6) / 1e9 for k, v in res.items() if k in ("fconv", "full", "mhsa", "tconvffn")}}
        print(dname, json.dumps(out[dname], indent=1))
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/time_kernels.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
