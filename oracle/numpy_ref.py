"""Independent numpy-fp64 restatement of the NetVLAD core, written from the closed-form
forward/backward formulas rather than through autograd.

TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see oracle/__init__.py).  Its purpose is to
cross-check the torch restatement in ``lpm_oracle.py`` (tier-1 self-consistency) and to pin
the exact algebra the HIP backward kernels implement (dU = u*dO - v*N with per-(clip,cluster)
coefficients) before any kernel exists.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-3
L2N_EPS = 1e-12


def softmax(z, axis=-1):
    z = z - z.max(axis=axis, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=axis, keepdims=True)


def netvlad_forward(x, W, gamma, beta, W2, B, T, residual=True, softmax_on=True, assign=None):
    """frame_level_models.py:2773-2824.  x [(B*T), D] -> dict with every saved quantity the
    backward needs.  With ``softmax_on=False`` ``assign`` [B,T,K] is used as given
    (NetVladAttenCluster, video_pooling_modules.py:1646-1658)."""
    D = x.shape[1]
    out = {}
    if softmax_on:
        L = x @ W                                             # :2781
        mu = L.mean(0)
        var = ((L - mu) ** 2).mean(0)                         # biased (slim.batch_norm train)
        rstd = 1.0 / np.sqrt(var + BN_EPS)
        Lhat = (L - mu) * rstd
        A = softmax(Lhat * gamma + beta, -1).reshape(B, T, -1)  # :2783-2801
        out.update(L=L, mu=mu, var=var, rstd=rstd, Lhat=Lhat)
    else:
        A = assign
    K = A.shape[-1]
    X = x.reshape(B, T, D)
    s = A.sum(1)                                              # [B,K]  :2803
    U = np.einsum("btk,btd->bdk", A, X)                       # :2812-2816
    if residual:
        U = U - s[:, None, :] * W2.reshape(1, D, K)           # :2805-2817
    n = (U * U).sum(1)                                        # [B,K] column square norms
    inv_n = 1.0 / np.sqrt(np.maximum(n, L2N_EPS))
    N = U * inv_n[:, None, :]                                 # :2819
    c = (N * N).sum(1)                                        # [B,K]
    g = c.sum(1)                                              # [B]
    inv_g = 1.0 / np.sqrt(np.maximum(g, L2N_EPS))
    O = (N * inv_g[:, None, None]).reshape(B, D * K)          # :2821-2822 (d-major)
    out.update(A=A, s=s, U=U, n=n, inv_n=inv_n, N=N, c=c, g=g, inv_g=inv_g, out=O)
    return out


def vlad_backward_coeffs(dO, N, W2, n, c, g):
    """Per-(clip, cluster) coefficients so that dU[b,:,k] = u[b,k]*dO[b,:,k] - v[b,k]*N[b,:,k].
    Derivation (tf.nn.l2_normalize with the max(.,eps) clamp, App. F.1):
       dN_k = inv_g (dO_k - m_g O_k alpha),  alpha = <dO,O>
       dU_k = inv_n_k (dN_k - m_k N_k <dN_k,N_k>)
    => dU_k = inv_n_k inv_g [ dO_k - N_k ( m_g alpha inv_g (1 - m_k c_k) + m_k p_k ) ],  p_k = <dO_k,N_k>.
    Also returns ctil[b,k] = sum_d dU[b,d,k] W2[d,k] (needed by dA, App. F.2)."""
    inv_n = 1.0 / np.sqrt(np.maximum(n, L2N_EPS))
    inv_g = 1.0 / np.sqrt(np.maximum(g, L2N_EPS))
    m_k = (n >= L2N_EPS).astype(np.float64)
    m_g = (g >= L2N_EPS).astype(np.float64)
    p = (dO * N).sum(1)                                       # [B,K]
    alpha = (p.sum(1) * inv_g)                                # <dO,O> = inv_g * sum_k p_k
    u = inv_n * inv_g[:, None]
    v = u * ((m_g * alpha * inv_g)[:, None] * (1.0 - m_k * c) + m_k * p)
    if W2 is not None:
        dow = (dO * W2[None]).sum(1)
        nw = (N * W2[None]).sum(1)
        ctil = u * dow - v * nw
    else:
        ctil = np.zeros_like(u)
    return u, v, ctil, p, alpha


def netvlad_backward(x, W, gamma, beta, W2, B, T, dOut, residual=True, softmax_on=True, assign=None):
    """Closed-form backward of netvlad_forward (App. F.1-F.4).  Returns grads for x, W, gamma,
    beta, W2 (and the assignment when softmax_on=False)."""
    f = netvlad_forward(x, W, gamma, beta, W2, B, T, residual, softmax_on, assign)
    D = x.shape[1]
    A, N = f["A"], f["N"]
    K = A.shape[-1]
    X = x.reshape(B, T, D)
    dO = dOut.reshape(B, D, K)
    W2m = W2.reshape(D, K) if residual else None
    u, v, ctil, _, _ = vlad_backward_coeffs(dO, N, W2m, f["n"], f["c"], f["g"])
    dU = u[:, None, :] * dO - v[:, None, :] * N                # [B,D,K]
    dA = np.einsum("bdk,btd->btk", dU, X) - ctil[:, None, :]   # F.2
    dX = np.einsum("btk,bdk->btd", A, dU)                      # F.2
    grads = {}
    if residual:
        grads["W2"] = -(f["s"][:, None, :] * dU).sum(0).reshape(W2.shape)
    if not softmax_on:
        grads["assign"] = dA
        grads["x"] = dX.reshape(B * T, D)
        return grads, f
    dLt = A * (dA - (A * dA).sum(-1, keepdims=True))           # F.3
    dLt = dLt.reshape(B * T, K)
    grads["gamma"] = (dLt * f["Lhat"]).sum(0)                  # F.4
    grads["beta"] = dLt.sum(0)
    M = B * T
    dL = gamma * f["rstd"] * (dLt - dLt.mean(0) - f["Lhat"] * (dLt * f["Lhat"]).mean(0))
    grads["W"] = x.T @ dL
    grads["x"] = dX.reshape(M, D) + dL @ W.T
    grads["dLt"] = dLt
    grads["dL"] = dL
    return grads, f


def attention_core(q, k, v, scale):
    """softmax(scale q k^T) v for [B,h,L,d] (transformer_utils.py:570-578)."""
    P = softmax(np.einsum("bhqd,bhkd->bhqk", q * scale, k), -1)
    return np.einsum("bhqk,bhkd->bhqd", P, v), P


def attention_core_backward(q, k, v, scale, dO):
    """App. F.6."""
    _, P = attention_core(q, k, v, scale)
    dV = np.einsum("bhqk,bhqd->bhkd", P, dO)
    dP = np.einsum("bhqd,bhkd->bhqk", dO, v)
    dS = P * (dP - (P * dP).sum(-1, keepdims=True))
    dQ = np.einsum("bhqk,bhkd->bhqd", dS, k) * scale
    dK = np.einsum("bhqk,bhqd->bhkd", dS, q) * scale
    return dQ, dK, dV


def moe_forward(act, Wg, We, be, vocab, m):
    """video_level_models.py:85-126."""
    gate = softmax((act @ Wg).reshape(-1, m + 1), -1)
    ex = 1.0 / (1.0 + np.exp(-(act @ We + be).reshape(-1, m)))
    return (gate[:, :m] * ex).sum(1).reshape(-1, vocab)


def cross_entropy(pred, labels):
    """losses.py:44-51."""
    eps = 10e-6
    y = labels.astype(np.float64)
    return float((-(y * np.log(pred + eps) + (1 - y) * np.log(1 - pred + eps))).sum(1).mean())


def adam_tf(p, g, m, v, lr, t, b1=0.9, b2=0.999, eps=1e-8):
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return p - lr_t * m / (np.sqrt(v) + eps), m, v
