"""The logits_bn attention core (NetVladV2's video stream: B = 80, L = 300, h = 64, d = 16) forward + backward alone, a few times: the
target of the PMC passes of tools/pmc_mha_bn.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops

dev = torch.device("cuda:0")
B, L, h, d = 80, 300, 64, 16
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, do = (torch.randn(B, L, h * d, device=dev, generator=g).requires_grad_(True) for _ in range(4))
gamma, beta = (1 + 0.1 * torch.randn(L, device=dev, generator=g)).requires_grad_(True), torch.zeros(L, device=dev, requires_grad=True)
mm, mv = torch.zeros(L, device=dev), torch.ones(L, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    o = ops.mha_core_bn(q, k, v, h, gamma, beta, mm, mv, is_training=True)
    o.backward(do)
torch.cuda.synchronize()
