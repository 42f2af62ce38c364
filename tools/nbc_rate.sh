cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_nbc_native.py tests/test_nb_models.py tests/test_abi.py tests/test_nbc2_native.py -m gpu -q -x 2>&1 | tail -8
python - <<'PY'
# NBC training step rate: BASELINE-like widths (192 / 8 heads / 384, 4 layers), batch 4 x 129 x 251, bf16-free fp32 stream: native vs torch.nn
import time, torch, warnings
from models.arch.NBC import NBC
import os
torch.manual_seed(0)
net = NBC(dim_input=16, dim_output=4, n_layers=4, encoder_kernel_size=4, n_heads=8, hidden_size=192, ffn_size=384).cuda().train()
x = torch.randn(4, 129, 251, 16, device="cuda")
def step():
    net.zero_grad(set_to_none=True)
    net(x).square().mean().backward()
for mode in ("1", "0"):
    os.environ["NBSS_NBC_NATIVE"] = mode
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(2): step()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): step()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"NBC train step batch 4: native={mode} {dt*1e3:.1f} ms ({4/dt:.1f} utt/s)")
PY
