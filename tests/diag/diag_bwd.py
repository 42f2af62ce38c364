import sys, torch
sys.path.insert(0, '.')  # run from the repo root: python tests/diag/diag_bwd.py
from nbss_amd._lib import hip, NBSS_BF16, NBSS_F32
from nbss_amd.engine import SpatialNetEngine
from oracle import spatialnet_ref as ref
lib = hip(); dev = torch.device('cuda:0')
L = 2
p = ref.init_params(num_layers=L, num_freqs=129, dim_input=12, dim_output=4, seed=0)
g = torch.Generator().manual_seed(1)
B, T = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 63
x = torch.randn(B, 129, T, 12, generator=g).bfloat16().to(dev)
dout = torch.randn(B, 129, T, 4, generator=g).to(dev)
def run(mode):
    eng = SpatialNetEngine(lib, dev, dim_input=12, dim_output=4, num_freqs=129, num_layers=L, dtype=NBSS_BF16)
    eng.load_params(p)
    eng.forward(x, train=True)
    eng.grads.zero_()
    if mode == 'whole':
        eng.backward(x, dout)
    else:
        eng.backward(x, dout, on_bucket=lambda lo, hi: None)
    torch.cuda.synchronize()
    return eng.param_views(eng.grads.clone()), eng
a, e = run('whole'); b, _ = run('whole'); c, _ = run('range')
for nm, (u, v) in (('whole-whole', (a, b)), ('whole-range', (a, c))):
    bad = []
    for k in u:
        d = float((u[k] - v[k]).norm() / (u[k].norm() + 1e-30))
        if d > 1e-6: bad.append((k, round(d, 6)))
    print(nm, 'differing:', bad[:40])
