"""Attention core fwd+bwd alone at the V1 video-encoder shape (B=80, L=256, h=64, d=16): rocprofv3 target."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
B, L, h, d = 80, 256, 64, 16
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, do = (torch.randn(B, L, h * d, device=dev, generator=g).requires_grad_(True) for _ in range(4))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    o = ops.mha_core(q, k, v, h, d ** -0.5)
    o.backward(do)
torch.cuda.synchronize()
print("ok")
