"""diagnostic: torch's native group_norm backward on the HIP device against the explicit form, by batch size and dtype"""
import torch
dev = torch.device("cuda:0")
for dt in (torch.float64, torch.float32, torch.bfloat16):
    for N in (64, 128, 129, 256, 1032):
        torch.manual_seed(0)
        h = torch.randn(N, 48, 16, device=dev, dtype=dt)
        r = torch.randn_like(h)
        res = []
        for explicit in (False, True):
            w = torch.randn(48, device=dev, dtype=dt, generator=torch.Generator(device=dev).manual_seed(1)).requires_grad_(True)
            b = torch.zeros(48, device=dev, dtype=dt, requires_grad=True)
            if explicit:
                hg = h.reshape(N, 8, -1)
                y = ((hg - hg.mean(-1, keepdim=True)) / torch.sqrt(hg.var(-1, unbiased=False, keepdim=True) + 1e-5)).reshape(N, 48, 16) * w[None, :, None] + b[None, :, None]
            else:
                y = torch.nn.functional.group_norm(h, 8, w, b, 1e-5)
            (y * r).sum().backward()
            res.append((w.grad.double(), b.grad.double()))
        print(dt, N, "dweight rel diff", float((res[0][0] - res[1][0]).norm() / res[1][0].norm()), "dbias", float((res[0][1] - res[1][1]).norm() / res[1][1].norm()))
