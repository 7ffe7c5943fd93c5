"""Runs only the NetVLAD forward kernels (K1, split, assign, K2, finalize) at BASELINE cfg-2 shapes a few times:
a short target for rocprofv3 --pmc passes (HBM traffic of the a5 chain, MFMA counters of K1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
B, T, D, K = 80, 300, 1024, 256
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B * T, 1152, device=dev, generator=g)
W = (torch.randn(D, K, device=dev, generator=g) / 32).requires_grad_(True)      # a gradient is wanted: the training-mode chain
W2 = torch.randn(1, D, K, device=dev, generator=g) / 32
bn = (torch.ones(K, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev), torch.ones(K, device=dev))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
LAZY = (sys.argv[2] if len(sys.argv) > 2 else "lazy") == "lazy"      # the production chain of NetVladV1: lazily normalised descriptor
for _ in range(n):
    out = ops.netvlad(x[:, :D], W, W2, T, bn=bn, kmajor=True, lazy=LAZY)
torch.cuda.synchronize()
print("ok", float(out.detach().norm()))
