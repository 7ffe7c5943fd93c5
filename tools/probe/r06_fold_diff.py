import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
R, N1, N2 = 128, 4160, 1024
g = torch.Generator().manual_seed(R + N2)
x = (torch.randn(R, N1, generator=g) / N1 ** 0.5).to(dev)
dy = torch.randn(R, N2, generator=g).to(dev)
p0 = (torch.randn(N1 * N2, generator=g) / 30).to(dev)
m0, v0 = (torch.randn(N1 * N2, generator=g) * 1e-3).to(dev), (torch.rand(N1 * N2, generator=g) * 1e-5).to(dev)
res = []
for fold in (False, True, True):
    fg = ops.FactoredGradient(); fg.put(x, dy)
    p, m, v = p0.clone(), m0.clone(), v0.clone()
    c16 = p0.view(N1, N2).to(torch.bfloat16).contiguous()
    dx = torch.full((R, N1), float("nan"), device=dev) if fold else None
    fg.clip_adam(p, m, v, 1.0, 2e-4, 3, param_bf16=c16, dx=dx)
    res.append((p, m, v, c16.float(), dx))
for k, name in enumerate(("p", "m", "v", "c16")):
    a, b, c = res[0][k], res[1][k], res[2][k]
    d = (a - b).abs()
    nz = (d > 0)
    print(name, "max abs diff", float(d.max()), "count", int(nz.sum()), "of", a.numel(), "rerun equal", bool(torch.equal(b, c)),
          "max rel", float((d / a.abs().clamp_min(1e-30)).max()))
    if int(nz.sum()):
        idx = nz.nonzero()[:8, 0]
        print("   first idx", [(int(i) // N2, int(i) % N2) for i in idx])
print("dx rerun equal", bool(torch.equal(res[1][4], res[2][4])))
