"""Attention core fwd / bwd alone at the V1 video-encoder shape (B=80, L=256, h=64, d=16), both arithmetics, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops

dev = torch.device("cuda:0")
B, L, h, d = 80, 256, 64, 16
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, do = (torch.randn(B, L, h * d, device=dev, generator=g).requires_grad_(True) for _ in range(4))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for prec in ("f32", "bf16x3"):
    ops.MHA_PRECISION = prec
    with torch.no_grad():
        tf = timeit(lambda: ops.mha_core(q, k, v, h, d ** -0.5))
    o = ops.mha_core(q, k, v, h, d ** -0.5)
    tb = timeit(lambda: o.backward(do, retain_graph=True))
    print(f"{prec:7s} fwd {tf:7.1f} us   bwd {tb:7.1f} us")
