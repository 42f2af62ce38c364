"""f4: utterances/s of the ON-DEVICE mixture simulator next to the training step it has to feed.
data_loaders/gpu_simulation.SimulatedRoomDataModule.batches at batch 32, 4-s, 6-channel, 2-speaker items (RIR FFT convolution, SIR / SNR
scaling, diffuse noise with the reference's coherence model): python tools/sim_throughput.py [batch] [batches]"""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from data_loaders.gpu_simulation import SimulatedRoomDataModule  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = "cuda:0" if torch.cuda.is_available() else "cpu"
    dm = SimulatedRoomDataModule(batch_size=[B, B], num_samples=[B * (nb + 2), B, B], audio_time_len=[4.0, 4.0, 4.0], device=dev)
    it = dm.batches(0)
    for _ in range(2):  # warm-up: rocFFT plans, allocator
        x, ys, _ = next(it)
    if dev != "cpu":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for x, ys, _ in it:
        n += x.shape[0]
    if dev != "cpu":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "SimulatedRoomDataModule.batches, stage 0", "device": dev, "batch": B, "utterances": n, "utt_per_s": n / dt,
                      "shape_x": list(x.shape), "shape_ys": list(ys.shape)}))


main()
