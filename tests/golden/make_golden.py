#!/usr/bin/env python3
"""Generates tests/golden/hotpath_golden.npz.

PROVENANCE: these vectors come from THIS REPO'S ORACLE (oracle/lpm_oracle.py, fp64), not from the reference:
the reference's hot path needs TensorFlow 1.x, which cannot be installed here, and the reference ships no
tests or golden vectors for it -- parity stays "unpinned" by the reference.  The fixture
freezes the oracle's answers on small seeded inputs so that (a) the oracle cannot drift silently and
(b) the GPU box can check the HIP path against committed data without trusting any code under oracle/.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lpm_oracle as O  # noqa: E402


def main():
    out = {}
    g = torch.Generator().manual_seed(2024)
    # --- NetVLAD forward/backward, training-mode cluster_bn (frame_level_models.py:2773-2824) ---
    B, T, D, K = 3, 20, 128, 16
    x = torch.randn(B * T, D, generator=g, dtype=torch.float64)
    W = torch.randn(D, K, generator=g, dtype=torch.float64) / D ** 0.5
    gamma = 1 + 0.3 * torch.randn(K, generator=g, dtype=torch.float64)
    beta = 0.2 * torch.randn(K, generator=g, dtype=torch.float64)
    W2 = torch.randn(1, D, K, generator=g, dtype=torch.float64) / D ** 0.5
    dout = torch.randn(B, D * K, generator=g, dtype=torch.float64)
    p = {"s/cluster_weights": W.clone().requires_grad_(True), "s/cluster_bn/gamma": gamma.clone().requires_grad_(True),
         "s/cluster_bn/beta": beta.clone().requires_grad_(True), "s/cluster_weights2": W2.clone().requires_grad_(True)}
    xr = x.clone().requires_grad_(True)
    y = O.netvlad_forward(xr, p, "s", T, True, True)
    y.backward(dout)
    out.update(nv_dims=np.array([B, T, D, K]), nv_x=x.numpy(), nv_W=W.numpy(), nv_gamma=gamma.numpy(), nv_beta=beta.numpy(),
               nv_W2=W2.numpy(), nv_dout=dout.numpy(), nv_out=y.detach().numpy(), nv_dx=xr.grad.numpy(),
               nv_dW=p["s/cluster_weights"].grad.numpy(), nv_dgamma=p["s/cluster_bn/gamma"].grad.numpy(),
               nv_dbeta=p["s/cluster_bn/beta"].grad.numpy(), nv_dW2=p["s/cluster_weights2"].grad.numpy())
    # --- attention core (transformer_utils.py:564-581) ---
    Bm, L, h, d = 2, 40, 4, 16
    q, k, v, do = (torch.randn(Bm, L, h * d, generator=g, dtype=torch.float64) for _ in range(4))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    o = O._combine_heads(O.attention_core(O._split_heads(qr, h), O._split_heads(kr, h), O._split_heads(vr, h), d ** -0.5))
    o.backward(do)
    out.update(mha_dims=np.array([Bm, L, h, d]), mha_q=q.numpy(), mha_k=k.numpy(), mha_v=v.numpy(), mha_do=do.numpy(),
               mha_o=o.detach().numpy(), mha_dq=qr.grad.numpy(), mha_dk=kr.grad.numpy(), mha_dv=vr.grad.numpy())
    # --- whole models, tiny: predictions of NetVladV1 / NetVladV2 (training-mode BN, V2 dropout off) ---
    for name, seed in (("NetVladV1", 11), ("NetVladV2", 12)):
        cfg = O.OracleConfig(model=name, iterations=12, cluster_size=16, hidden_size=32, vocab_size=40, v2_dropout_rate=0.0)
        xin, nf, _ = O.make_synthetic_batch(3, 16, 1152, 40, seed=seed, min_frames=6)
        prm = {n: t.double() for n, t in O.init_params(cfg, 1152, seed=1000 + seed).items()}
        pred = O.model_forward(prm, xin.double(), nf, cfg, True)
        out[f"{name}_input"] = xin.numpy()
        out[f"{name}_num_frames"] = nf.numpy()
        out[f"{name}_pred"] = pred.numpy()
        out[f"{name}_param_seed"] = np.array([1000 + seed])
    path = os.path.join(ROOT, "tests", "golden", "hotpath_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
