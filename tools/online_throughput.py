"""f2 (BASELINE config 5): frames/s of the OnlineSpatialNet streaming step — the native HIP step
(nbss_amd/online.py, HIP graph per chunk) next to the torch.nn step under the same graph capture (OnlineStreamer) and the eager
torch.nn step.  configs/onlineSpatialNet.yaml geometry (8 layers, 129 frequencies, 6 channels -> 2 speakers, ret(2)):
    python tools/online_throughput.py [chunk] [seconds] [attention, e.g. "mhsa(251)"]"""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from models.arch.OnlineSpatialNet import OnlineSpatialNet, OnlineStreamer  # noqa: E402
from nbss_amd.online import NativeOnlineStreamer  # noqa: E402


def main():
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 32.0
    attention = sys.argv[3] if len(sys.argv) > 3 else "ret(2)"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = OnlineSpatialNet(dim_input=12, dim_output=4, num_layers=8, dim_squeeze=8, num_freqs=129, encoder_kernel_size=5, dim_hidden=96, dim_ffn=192,
                           num_heads=4, dropout=(0, 0, 0), kernel_size=(5, 3), conv_groups=(8, 8), norms=["LN", "LN", "GN", "LN", "LN", "LN"], full_share=0,
                           attention=attention, decay=[4, 5, 9, 10], rope=False).eval().to(dev)
    T = int(secs * 8000) // 128 // chunk * chunk
    x = torch.randn(1, 129, T, 12, device=dev)
    out = {"what": f"OnlineSpatialNet {attention} streaming step, batch 1, 129 frequencies, 8 layers", "chunk_frames": chunk, "frames": T,
           "audio_seconds": T * 128 / 8000}
    ref = None
    for name, mk in (("native_hip_graph", lambda: NativeOnlineStreamer(net, 1, chunk, device=dev, use_graph=True)),
                     ("torch_nn_graph", lambda: OnlineStreamer(net, 1, chunk, device=dev, use_graph=True)),
                     ("torch_nn_eager", lambda: OnlineStreamer(net, 1, chunk, device=dev, use_graph=False))):
        s = mk()
        s.step(x[:, :, :chunk])  # capture / warm-up
        s.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ys = [s.step(x[:, :, c:c + chunk]) for c in range(0, T, chunk)]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        y = torch.cat(ys, 2)
        if ref is None:
            ref = y
        out[name] = {"frames_per_s": T / dt, "ms_per_chunk": dt / (T // chunk) * 1e3, "real_time_factor": (T * 128 / 8000) / dt,
                     "rel_l2_vs_native": float((y - ref).norm() / ref.norm())}
    print(json.dumps(out))


main()
