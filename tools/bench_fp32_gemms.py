"""Time the fp32 library GEMMs left in the cfg-2 step (skinny M = batch 80 and the K1 backward) with HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
B, H, V = 80, 512, 3862


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def r(*s):
    return torch.randn(*s, device=dev)


cases = []
x, dy = r(B, H), None
for name, N in (("moe gates", V * 3), ("moe experts", V * 2), ("gating", H)):
    W, dyn = r(H, N), r(B, N)
    cases.append((f"{name} fwd [80,512]x[512,{N}]", lambda x=x, W=W: x.matmul(W), 4 * H * N))
    cases.append((f"{name} dx  [80,{N}]x[{N},512]", lambda dyn=dyn, W=W: dyn.matmul(W.t()), 4 * H * N))
    cases.append((f"{name} dW  [512,80]x[80,{N}]", lambda x=x, dyn=dyn: x.t().matmul(dyn), 4 * H * N))
KV = 270336
xp, Wp, dyp = r(B, KV), r(KV, H), r(B, H)
out = torch.empty(KV, H, device=dev)
cases.append(("proj fwd split-K bmm", lambda: torch.bmm(xp.view(B, 132, KV // 132).transpose(0, 1), Wp.view(132, KV // 132, H)).sum(0), 4 * KV * H))
cases.append(("proj dx [80,512]x[512,270336]", lambda: dyp.matmul(Wp.t()), 4 * KV * H))
cases.append(("proj dW [270336,80]x[80,512] out=", lambda: torch.mm(xp.t(), dyp, out=out), 4 * KV * H))
M = 24000
for D, K in ((1024, 256), (128, 64)):
    xx, dl, W, dx = r(M, 1152)[:, :D], r(M, K), r(D, K), r(M, D)
    cases.append((f"K1 bwd dW [{D},{M}]x[{M},{K}]", lambda xx=xx, dl=dl: xx.t().matmul(dl), 4 * M * (D + K)))
    cases.append((f"K1 bwd dx += [{M},{K}]x[{K},{D}]", lambda dx=dx, dl=dl, W=W: dx.addmm_(dl, W.t()), 4 * M * (2 * D + K)))
for name, fn, byts in cases:
    t = timeit(fn)
    print(f"{name:45s} {t:8.1f} us   HBM floor {byts / 6.3e6:7.1f} us")
