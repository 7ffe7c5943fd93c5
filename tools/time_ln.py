"""Residual layer norm (fwd + bwd through ops.residual_layer_norm) alone at the video encoder's shape [80, 256, 1024]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
B, L, F = 80, 256, 1024
a = torch.randn(B, L, F, device=dev, requires_grad=True)
r = torch.randn(B, L, F, device=dev, requires_grad=True)
g = torch.ones(F, device=dev, requires_grad=True)
be = torch.zeros(F, device=dev, requires_grad=True)
dy = torch.randn(B, L, F, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    y = ops.residual_layer_norm(a, r, g, be)
    y.backward(dy)
torch.cuda.synchronize()
print("done")
