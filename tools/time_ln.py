"""Residual layer norm (fwd + bwd through ops.residual_layer_norm) alone at the video encoder's shape [80, 256, 1024].
argv[2] = 1: a 1 GB fill between the passes pushes the tensors out of the last-level cache (cold-data timing)."""
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
cold = len(sys.argv) > 2 and sys.argv[2] == "1"
junk = torch.empty(256 * 1024 * 1024, device=dev) if cold else None
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    if cold:
        junk.fill_(1.0)
    y = ops.residual_layer_norm(a, r, g, be)
    if cold:
        junk.fill_(2.0)
    y.backward(dy)
torch.cuda.synchronize()
print("done")
