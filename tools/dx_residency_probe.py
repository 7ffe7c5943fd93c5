"""After the projection's input gradient dx [80, 270336] has been written -- by the own weight-stream kernel or by the library's fp32 GEMM --
how long does the NEXT reader of dx take (a plain sum over it, 86 MB)?  Events around the reader only; 30 rounds, alternating.
  python tools/dx_residency_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
M, Kd, N = 80, 270336, 512
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, Kd, device=dev, generator=g).requires_grad_(True)
W = (torch.randn(Kd, N, device=dev, generator=g) / 16).requires_grad_(True)
dy = torch.randn(M, N, device=dev, generator=g)
other = torch.randn(M, Kd, device=dev, generator=g)          # a second 86 MB tensor read right after (the layer norm reads dy AND z)
res = {True: [], False: []}
for rnd in range(40):
    for own in (True, False):
        ops.PROJ_DX_STREAM_MIN_N = 512 if own else 1 << 30
        y = ops.projection(x, W)
        dxv, = torch.autograd.grad(y, x, dy)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s = dxv.sum() + other.sum()
        e1.record()
        torch.cuda.synchronize()
        if rnd >= 5:
            res[own].append(e0.elapsed_time(e1) * 1e3)
for own in (True, False):
    v = sorted(res[own])
    print(f"{'own kernel' if own else 'library GEMM'}: the reader after dx takes median {v[len(v) // 2]:.1f} us (min {v[0]:.1f}, max {v[-1]:.1f})")
