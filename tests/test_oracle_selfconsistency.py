"""Tier 1: torch restatement vs independent numpy closed forms, fp64."""
import numpy as np
import torch

from oracle import lpm_oracle as O
from oracle import numpy_ref as R


def _mk(B=3, T=7, D=12, K=5, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B * T, D))
    W = rng.standard_normal((D, K)) / np.sqrt(D)
    gamma = 1.0 + 0.3 * rng.standard_normal(K)
    beta = 0.2 * rng.standard_normal(K)
    W2 = rng.standard_normal((1, D, K)) / np.sqrt(D)
    dOut = rng.standard_normal((B, D * K))
    return x, W, gamma, beta, W2, dOut


def _params(W, gamma, beta, W2):
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    return {"s/cluster_weights": t(W), "s/cluster_bn/gamma": t(gamma), "s/cluster_bn/beta": t(beta),
            "s/cluster_weights2": t(W2)}


def test_netvlad_forward_matches_numpy():
    B, T = 3, 7
    x, W, gamma, beta, W2, _ = _mk(B, T)
    p = _params(W, gamma, beta, W2)
    out = O.netvlad_forward(torch.tensor(x), p, "s", T, True, True)
    ref = R.netvlad_forward(x, W, gamma, beta, W2, B, T)["out"]
    np.testing.assert_allclose(out.detach().numpy(), ref, rtol=1e-10, atol=1e-12)


def test_netvlad_backward_closed_form_matches_autograd():
    B, T = 3, 7
    x, W, gamma, beta, W2, dOut = _mk(B, T)
    p = _params(W, gamma, beta, W2)
    xt = torch.tensor(x, requires_grad=True)
    out = O.netvlad_forward(xt, p, "s", T, True, True)
    out.backward(torch.tensor(dOut))
    g, _ = R.netvlad_backward(x, W, gamma, beta, W2, B, T, dOut)
    np.testing.assert_allclose(xt.grad.numpy(), g["x"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(p["s/cluster_weights"].grad.numpy(), g["W"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(p["s/cluster_bn/gamma"].grad.numpy(), g["gamma"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(p["s/cluster_bn/beta"].grad.numpy(), g["beta"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(p["s/cluster_weights2"].grad.numpy(), g["W2"], rtol=1e-8, atol=1e-10)


def test_backward_with_degenerate_column():
    """A cluster whose residual column is exactly zero exercises the max(.,1e-12) clamp."""
    B, T, D, K = 2, 4, 6, 3
    rng = np.random.default_rng(3)
    sims = rng.standard_normal((B, T, K))
    sims[:, :, 1] = 0.0                      # cluster 1 gets no mass -> U[:, :, 1] == 0
    x = rng.standard_normal((B * T, D))
    C = rng.standard_normal((D, K))
    dOut = rng.standard_normal((B, D * K))
    st = torch.tensor(sims, requires_grad=True)
    xt = torch.tensor(x, requires_grad=True)
    Ct = torch.tensor(C, requires_grad=True)
    out = O.vlad_aggregate(st, xt.reshape(B, T, D), Ct)
    out.backward(torch.tensor(dOut))
    g, _ = R.netvlad_backward(x, None, None, None, C, B, T, dOut, softmax_on=False, assign=sims)
    np.testing.assert_allclose(st.grad.numpy(), g["assign"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(xt.grad.numpy(), g["x"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(Ct.grad.numpy(), g["W2"], rtol=1e-7, atol=1e-9)


def test_v2_explicit_4d_equals_gemm_form():
    cfg = O.OracleConfig(model="NetVladV2", iterations=6, cluster_size=8, hidden_size=16, vocab_size=10)
    p = {k: v.double() for k, v in O.init_params(cfg, 1152, seed=5).items()}
    x = torch.randn(2 * 6, 128, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    a = O.netvlad_atten_cluster_forward(x, p, "audio_VLAD", 6, False)
    b = O.netvlad_atten_cluster_forward(x, p, "audio_VLAD", 6, False, explicit_4d=True)
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-10, atol=1e-12)


def test_attention_core_backward_closed_form():
    rng = np.random.default_rng(7)
    q, k, v, dO = (rng.standard_normal((2, 3, 5, 4)) for _ in range(4))
    tq, tk, tv = (torch.tensor(a, requires_grad=True) for a in (q, k, v))
    o = O.attention_core(tq, tk, tv, 0.5)
    o.backward(torch.tensor(dO))
    dQ, dK, dV = R.attention_core_backward(q, k, v, 0.5, dO)
    np.testing.assert_allclose(tq.grad.numpy(), dQ, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(tk.grad.numpy(), dK, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(tv.grad.numpy(), dV, rtol=1e-9, atol=1e-11)


def test_fp32_vs_fp64_full_model():
    cfg = O.OracleConfig(model="NetVladV1", iterations=10, cluster_size=8, hidden_size=32, vocab_size=50)
    x, nf, lab = O.make_synthetic_batch(4, 12, 1152, 50, seed=0)
    p32 = O.init_params(cfg, 1152, seed=1000)
    p64 = {k: v.double() for k, v in p32.items()}
    a = O.model_forward(p32, x, nf, cfg, True)
    b = O.model_forward(p64, x.double(), nf, cfg, True)
    assert (a.double() - b).abs().max().item() < 1e-5
