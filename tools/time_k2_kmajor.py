"""K2 (vlad_aggregate_tiles3) alone at cfg-2, durations from launch-attached HIP events."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops, _capi
dev = torch.device("cuda:0")
B, T, D, K = 80, 300, 1024, 256
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B * T, 1152, device=dev, generator=g)
W = torch.randn(D, K, device=dev, generator=g) / 32
W2 = torch.randn(1, D, K, device=dev, generator=g) / 32
bn = (torch.ones(K, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev), torch.ones(K, device=dev))
lib = _capi.load()
with torch.no_grad():
    for _ in range(3):
        ops.netvlad(x[:, :D], W, W2, T, bn=bn, kmajor=True)
    lib._lpm_kernel_timing_enable(1)
    for _ in range(20):
        out = ops.netvlad(x[:, :D], W, W2, T, bn=bn, kmajor=True)
    torch.cuda.synchronize()
    lib._lpm_kernel_timing_enable(0)
buf = (ctypes.c_float * 256)()
n = lib._lpm_kernel_timing_read(2, buf, 256)
ms = sorted(buf[i] for i in range(n))
print("K2 launches %d  median %.1f us  min %.1f us   (207.8 MB algorithmic -> %.2f TB/s at the median)" % (n, ms[n // 2] * 1e3, ms[0] * 1e3, 207.8e6 / (ms[n // 2] * 1e-3) / 1e12))
