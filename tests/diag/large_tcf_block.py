"""diagnostic: the generic T-ConvFFN backward alone (large geometry) at (B, F, T): GroupNorm affine gradients against the oracle"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from nbss_amd import ops  # noqa: E402
from nbss_amd._lib import NBSS_F32, hip, make_cfg  # noqa: E402
from nbss_amd.params import param_table  # noqa: E402
from oracle import spatialnet_ref as ref  # noqa: E402
from util import rel_l2  # noqa: E402
B, F, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
lib = hip()
kw = dict(dim_hidden=192, dim_ffn=384, dim_squeeze=16)
cfg = make_cfg(B, F, T, 12, 4, L=1, dtype=NBSS_F32, H=192, FFN=384, SQ=16)
p = ref.init_params(num_layers=1, num_freqs=F, seed=7, **kw)
flat = ops.flatten_params(lib, cfg, p, dev)
packed = ops.pack_params(lib, cfg, flat)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, F, T, 192, generator=g).to(dev)
dy = torch.randn(B, F, T, 192, generator=g).to(dev)
G = torch.zeros_like(flat)
ws = ops.workspace(lib, cfg, dev)
dx = ops.tconvffn_bwd(lib, cfg, flat, G, packed, 0, x, dy, ws)
torch.cuda.synchronize()
table = param_table(lib, cfg)
x64 = x.double().requires_grad_(True)
p64 = {k: v.double().to(dev).requires_grad_(True) for k, v in p.items()}
y = ref.tconvffn(x64, p64, "layers.0")
(y * dy.double()).sum().backward()
for n in ["layers.0.tconvffn.6.weight", "layers.0.tconvffn.6.bias", "layers.0.tconvffn.5.bias", "layers.0.tconvffn.8.weight"]:
    off, shape = table[n]
    got = G[off:off + p64[n].numel()].reshape(shape)
    print(n, rel_l2(got, p64[n].grad))
print("dx", rel_l2(dx, x64.grad))
# which sequences made it into the GroupNorm weight gradient?  per-sequence contributions from the oracle, least squares for their weights
n = "layers.0.tconvffn.6.weight"
off, shape = table[n]
got = G[off:off + 384].double()
cs = []
for s in range(B * F):
    xs_ = x.reshape(B * F, T, 192)[s:s + 1].reshape(1, 1, T, 192).double().requires_grad_(True)
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in p64.items()}
    ys = ref.tconvffn(xs_, ps, "layers.0")
    (ys * dy.reshape(B * F, T, 192)[s:s + 1].reshape(1, 1, T, 192).double()).sum().backward()
    cs.append(ps[n].grad.reshape(-1))
Cm = torch.stack(cs, 1)  # [384][nseq]
w = torch.linalg.lstsq(Cm, got.reshape(-1, 1)).solution.reshape(-1)
print("weights of the sequences:", [round(float(v), 2) for v in w])
