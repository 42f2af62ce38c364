cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_blstm_native.py tests/test_nb_models.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -6
python - <<'PY'
# NB-BLSTM (reference configs/NB-BLSTM.yaml: hidden 256 / 128, fp32) training step, batch 4 x 129 x 251: native vs torch.nn (MIOpen LSTM)
import os, time, torch, warnings
from models.arch.blstm2_fc1 import BLSTM2_FC1
torch.manual_seed(0)
net = BLSTM2_FC1(dim_input=12, dim_output=4, hidden_size=(256, 128)).cuda().train()
x = torch.randn(4, 129, 251, 12, device="cuda")
def step():
    net.zero_grad(set_to_none=True)
    net(x).square().mean().backward()
for mode in ("1", "0"):
    os.environ["NBSS_BLSTM_NATIVE"] = mode
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(2): step()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): step()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        with torch.no_grad():
            net.eval(); net(x); torch.cuda.synchronize(); t0 = time.time()
            for _ in range(5): net(x)
            torch.cuda.synchronize(); di = (time.time() - t0) / 5
            net.train()
    print(f"NB-BLSTM batch 4: native={mode} train step {dt*1e3:.1f} ms ({4/dt:.1f} utt/s), inference {di*1e3:.1f} ms")
PY
