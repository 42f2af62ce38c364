import torch, time, sys
sys.path.insert(0, '.')  # run from the repo root: python tests/diag/long_time.py
from nbss_amd._lib import hip as hip_lib, NBSS_BF16, NBSS_F32
from nbss_amd.engine import SpatialNetEngine
from oracle import spatialnet_ref as ref
lib = hip_lib()
dev = torch.device('cuda:0')
for dtype, nm in ((NBSS_BF16, 'bf16'), (NBSS_F32, 'f32')):
    eng = SpatialNetEngine(lib, dev, dim_input=12, dim_output=4, num_freqs=129, num_layers=8, dtype=dtype)
    eng.load_params(ref.init_params(num_layers=8, num_freqs=129, dim_input=12, dim_output=4, seed=4))
    for (B, T) in ((1, 251), (1, 600), (1, 1001), (4, 1001), (1, 2000)):
        x = torch.randn(B, 129, T, 12, device=dev).to(eng.stream_dtype())
        for _ in range(2): eng.forward(x, train=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): eng.forward(x, train=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{nm} B={B} T={T}: {dt*1e3:.2f} ms  ({dt*1e3/(B*T/62.5):.3f} ms per audio-second)", flush=True)
