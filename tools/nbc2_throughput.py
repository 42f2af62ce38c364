"""NBC2 (BASELINE config 4's network: 8 layers, 96 / 192, 2 heads, 8 ch -> 3 spk, fp32) on the device: the native paths (nbss_amd/nbc2.py over the
nbss_nb_* building blocks: inference forward, and the training forward + backward) against the torch.nn modules (ATen / MIOpen kernels).
usage: python tools/nbc2_throughput.py [batch] [reps]"""
import json
import sys
import time
import warnings
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import models.arch.NBC2 as M  # noqa: E402
from models.arch.NBC2 import NBC2  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = NBC2(dim_input=16, dim_output=6, n_layers=8, dim_hidden=96, dim_ffn=192, num_freqs=129).to(dev)
    x = torch.randn(B, 129, 251, 16, device=dev)
    r = torch.randn(B, 129, 251, 6, device=dev)
    out = {}
    runner = net._native()
    assert runner is not None

    def use_native(on):
        M._NATIVE[net] = (runner if on else None, None if on else "switched off by tools/nbc2_throughput.py")

    # inference
    with torch.no_grad():
        use_native(True)
        y_native = net(x)
        use_native(False)
        y_torch = net(x)
        out["rel_l2_native_vs_torch"] = float((y_native - y_torch).norm() / y_torch.norm())
        for name, on in (("native", True), ("torch_nn", False)):
            use_native(on)
            dt = timed(lambda: net(x), reps)
            out[name] = {"ms_per_batch": round(dt * 1e3, 2), "utt_per_s": round(B / dt, 1)}

    # training: forward + backward of sum(y * r) (every parameter gradient), no optimizer
    def step():
        net.zero_grad(set_to_none=True)
        (net(x) * r).sum().backward()

    grads = {}
    for name, on in (("train_native", True), ("train_torch_nn", False)):
        use_native(on)
        dt = timed(step, reps)
        out[name] = {"ms_per_step": round(dt * 1e3, 2), "utt_per_s": round(B / dt, 1)}
        grads[name] = torch.cat([p.grad.reshape(-1).double() for p in net.parameters()])
    out["rel_l2_grads_native_vs_torch"] = float((grads["train_native"] - grads["train_torch_nn"]).norm() / grads["train_torch_nn"].norm())
    print(json.dumps({"what": "NBC2, 8 layers, 129 x 251, 8 ch -> 3 spk, fp32: inference forward and training forward + backward", "batch": B, **out}))


if __name__ == "__main__":
    warnings.simplefilter("ignore")  # (the module reports its torch.nn path once per reason)
    main()
