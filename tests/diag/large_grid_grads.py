"""diagnostic: SpatialNet-large train step at a given grid, every parameter gradient against the fp64 oracle evaluated on the device
usage: python tests/diag/large_grid_grads.py [T] [L] [dtype f32|bf16] [F] [B]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from nbss_amd._lib import NBSS_BF16, NBSS_F32, hip  # noqa: E402
from nbss_amd.engine import SpatialNetEngine  # noqa: E402
from oracle import spatialnet_ref as ref  # noqa: E402
from util import rel_l2  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 251
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dt = NBSS_F32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else NBSS_BF16
F = int(sys.argv[4]) if len(sys.argv) > 4 else 129
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda:0")
kw = dict(dim_hidden=192, dim_ffn=384, dim_squeeze=16)
p = ref.init_params(num_layers=L, num_freqs=F, seed=7, **kw)
g = torch.Generator().manual_seed(8)
x = torch.randn(B, F, T, 12, generator=g)
r = torch.randn(B, F, T, 4, generator=g)
eng = SpatialNetEngine(hip(), dev, dim_input=12, dim_output=4, num_freqs=F, num_layers=L, dtype=dt, **kw)
eng.load_params(p)
xs = x.to(eng.stream_dtype()).to(dev)
y = eng.forward(xs, train=True)
eng.grads.zero_()
eng.backward(xs, r.to(dev))
views = eng.param_views(eng.grads)
leaves = {}
p64 = {k: leaves.setdefault(id(v), v.double().to(dev).requires_grad_(True)) for k, v in p.items()}
wy = ref.spatialnet(xs.double(), p64, L)
(wy * r.double().to(dev)).sum().backward()
print("y", rel_l2(y, wy.detach()))
errs = sorted(((rel_l2(views[k], v.grad), k) for k, v in p64.items()), reverse=True)
for e, k in errs[:14]:
    print(f"{e:.4e} {k}")
e, k = errs[0]
gq, wq = views[k].double().cpu().reshape(-1), p64[k].grad.cpu().reshape(-1)
print(k, "got", [round(float(v), 4) for v in gq[:10]], "\nwant", [round(float(v), 4) for v in wq[:10]], "\nratio", [round(float(a / b), 3) for a, b in zip(gq[:10], wq[:10])])
print("ratio by group:", [round(float(gq[i * 48:(i + 1) * 48].norm() / wq[i * 48:(i + 1) * 48].norm()), 3) for i in range(8)])
