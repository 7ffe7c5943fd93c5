"""bf16 storage at a ragged clip length: where does the product leave the oracle?  (round 6 debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import lpm_oracle as O
from learnablepoolingmethods_amd import ops

dev = torch.device("cuda:0")
for (B, T, K) in ((4, 300, 512), (3, 77, 512), (3, 128, 512), (3, 64, 512), (3, 100, 512), (2, 300, 256)):
    D = 1024
    g = torch.Generator().manual_seed(B * 7 + T)
    raw = torch.randn(B, T, 1152, generator=g)
    nf = torch.full((B,), T, dtype=torch.int32)
    W = torch.randn(D, K, generator=g) / D ** 0.5
    gamma, beta = 1 + 0.3 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    W2 = torch.randn(1, D, K, generator=g) / D ** 0.5
    x = raw.reshape(B * T, 1152)[:, :D]
    p = {"s/cluster_weights": W.double(), "s/cluster_bn/gamma": gamma.double(), "s/cluster_bn/beta": beta.double(), "s/cluster_weights2": W2.double()}
    ref = O.netvlad_forward(x.double(), p, "s", T, True, True, {})
    for form in (True, False):
        ops.VLAD_CLIP16 = form
        y = ops.frame_sample_bn(raw.to(dev), nf.to(dev), T, storage="bf16", materialize=False)
        with torch.no_grad():
            xs = y[:, :D]
        Wg, gmg, btg, W2g = (t.to(dev).requires_grad_(True) for t in (W, gamma, beta, W2))
        out = ops.netvlad(xs, Wg, W2g, T, bn=(gmg, btg, torch.zeros(K, device=dev), torch.ones(K, device=dev)), is_training=True, storage="bf16")
        sv = out.grad_fn.saved_tensors
        logits = sv[2].float().cpu().double()
        lref = x.double() @ W.double()
        e_l = float((logits - lref).abs().max() / lref.abs().max())
        e_o = float((out.float().cpu().double() - ref).abs().max() / ref.abs().max())
        asum = sv[10].cpu().double()
        print(f"B={B} T={T} K={K} clip16={form}: logits {e_l:.2e}, descriptor {e_o:.2e}, asum total {float(asum.sum()):.3f} (expect {B * T})", flush=True)
