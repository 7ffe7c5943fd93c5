"""Prints the relative error of both K2 arithmetics against the fp64 oracle at cfg-2 layer sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import lpm_oracle as O
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
B, T, D, K = 4, 300, 1024, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(B * T, 1152, generator=g); W = torch.randn(D, K, generator=g) / 32
gamma = 1 + 0.3 * torch.randn(K, generator=g); beta = 0.2 * torch.randn(K, generator=g); W2 = torch.randn(1, D, K, generator=g) / 32
p = {"s/cluster_weights": W.double(), "s/cluster_bn/gamma": gamma.double(), "s/cluster_bn/beta": beta.double(), "s/cluster_weights2": W2.double()}
ref = O.netvlad_forward(x[:, :D].double(), p, "s", T, True, True)
for prec in ("bf16x3", "f32"):
    ops.VLAD_PRECISION = prec
    out = ops.netvlad(x.to(dev)[:, :D], W.to(dev), W2.to(dev), T, bn=(gamma.to(dev), beta.to(dev), torch.zeros(K, device=dev), torch.ones(K, device=dev)))
    d = (out.double().cpu() - ref)
    print(prec, "max|err|/max|ref| = %.3e" % float(d.abs().max() / ref.abs().max()), " rel L2 = %.3e" % float(d.norm() / ref.norm()),
          " max elementwise rel (|ref|>1e-4) = %.3e" % float((d.abs() / ref.abs().clamp_min(1e-4)).max()))
