"""Independent numpy-fp64 restatement of the hot path -- the NetVLAD core, and since round 3 also joint layer norm, rank-2/3/4
batch norm, both encoder blocks (MultiHeadAttention / MultiHeadAttentionBN + their feed-forward networks), NetVladV2's aggregator
and clip / combine / learning rate -- written from the closed-form forward/backward formulas (SURVEY.md App. A.4 / App. F and the
reference sources) rather than through autograd.

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


# ----------------------------------------------------------------------------------------------------------------------
# Round 3: the rest of the hot path, restated independently of lpm_oracle.py -- forward AND hand-derived backward, so that a
# misreading shared by the torch restatement (which leans on autograd) and the kernels has a second, differently built witness.
# Written from the reference sources: transformer_utils.py:374-457 (encoders), :507-586 (MultiHeadAttention), :589-677
# (MultiHeadAttentionBN), :679-766 (feed-forward networks), video_pooling_modules.py:1592-1663 (NetVladAttenCluster),
# utils.py:170-213 (clip / combine), train.py:244-252 (learning rate) and the TF1 op semantics of SURVEY App. B.
# Parameters are dicts keyed by the variable names TF1 gives them inside the module's scope ("q/kernel", "LayerNorm/gamma", ...).
# ----------------------------------------------------------------------------------------------------------------------
LN_EPS = 1e-12


def layer_norm_fwd(x, gamma, beta):
    """tf.contrib.layers.layer_norm defaults (begin_norm_axis = 1): ONE mean / variance per example over every non-batch axis,
    gamma / beta on the last axis, eps 1e-12 (transformer_utils.py:407,411,454,713)."""
    ax = tuple(range(1, x.ndim))
    mu = x.mean(ax, keepdims=True)
    var = ((x - mu) ** 2).mean(ax, keepdims=True)
    rstd = 1.0 / np.sqrt(var + LN_EPS)
    xh = (x - mu) * rstd
    return xh * gamma + beta, (xh, rstd, gamma)


def layer_norm_bwd(dy, cache):
    xh, rstd, gamma = cache
    ax = tuple(range(1, dy.ndim))
    red = tuple(range(dy.ndim - 1))
    g = dy * gamma
    dx = rstd * (g - g.mean(ax, keepdims=True) - xh * (g * xh).mean(ax, keepdims=True))
    return dx, (dy * xh).sum(red), dy.sum(red)


def batch_norm_fwd(x, gamma, beta):
    """slim.batch_norm(center, scale, is_training=True): channel = LAST axis, statistics over all the others, biased variance in the
    normalisation, eps 1e-3.  Third result: what the moving averages are fed -- the fused kernel (rank 2 / 4 inputs) hands over the
    UNBIASED variance, the generic path (rank 3) the biased one."""
    red = tuple(range(x.ndim - 1))
    mu = x.mean(red)
    var = ((x - mu) ** 2).mean(red)
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    xh = (x - mu) * rstd
    n = x.size // x.shape[-1]
    fed = var * (n / max(n - 1, 1)) if x.ndim in (2, 4) else var
    return xh * gamma + beta, (xh, rstd, gamma), {"moving_mean": mu, "moving_variance": fed}


def batch_norm_bwd(dy, cache):
    xh, rstd, gamma = cache
    red = tuple(range(dy.ndim - 1))
    g = dy * gamma
    dx = rstd * (g - g.mean(red) - xh * (g * xh).mean(red))
    return dx, (dy * xh).sum(red), dy.sum(red)


def _split(x, h):                 # [B, L, F] -> [B, h, L, F/h]   (transformer_utils.py:521-540)
    B, L, F = x.shape
    return x.reshape(B, L, h, F // h).transpose(0, 2, 1, 3)


def _combine(x):                  # [B, h, L, d] -> [B, L, h d]   (:542-550)
    B, h, L, d = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, L, h * d)


def mha_fwd(x, p, h):
    """MultiHeadAttention.forward(x, x) (transformer_utils.py:552-586): bias-free q / k / v, q scaled by depth^-0.5, softmax, .v,
    combine, dense + bias."""
    F = x.shape[-1]
    q, k, v = _split(x @ p["q/kernel"], h), _split(x @ p["k/kernel"], h), _split(x @ p["v/kernel"], h)
    scale = (F // h) ** -0.5
    o, _ = attention_core(q, k, v, scale)
    oc = _combine(o)
    return oc @ p["output_transform/kernel"] + p["output_transform/bias"], (x, q, k, v, scale, oc, h)


def mha_bwd(dy, cache, p):
    x, q, k, v, scale, oc, h = cache
    F = x.shape[-1]
    g = {"output_transform/kernel": oc.reshape(-1, F).T @ dy.reshape(-1, dy.shape[-1]), "output_transform/bias": dy.sum((0, 1))}
    do = _split(dy @ p["output_transform/kernel"].T, h)
    dq, dk, dv = attention_core_backward(q, k, v, scale, do)
    dq, dk, dv = _combine(dq), _combine(dk), _combine(dv)
    x2 = x.reshape(-1, F)
    g["q/kernel"], g["k/kernel"], g["v/kernel"] = x2.T @ dq.reshape(-1, F), x2.T @ dk.reshape(-1, F), x2.T @ dv.reshape(-1, F)
    dx = dq @ p["q/kernel"].T + dk @ p["k/kernel"].T + dv @ p["v/kernel"].T
    return dx, g


def mha_bn_fwd(x, p, h):
    """MultiHeadAttentionBN.forward(x, x) (transformer_utils.py:634-677): NO q scaling; slim.batch_norm on the rank-4 logits
    [B, h, Lq, Lk] -- channel = key position, statistics over (B, h, Lq); softmax; .v; combine; slim.batch_norm over the hidden axis
    of the rank-3 result; dense + bias.  Also returns the moving-average feeds of the two batch norms."""
    q, k, v = _split(x @ p["q/kernel"], h), _split(x @ p["k/kernel"], h), _split(x @ p["v/kernel"], h)
    logits = np.einsum("bhqd,bhkd->bhqk", q, k)
    lb, c_lbn, f1 = batch_norm_fwd(logits, p["logits_bn/gamma"], p["logits_bn/beta"])
    P = softmax(lb, -1)
    oc = _combine(np.einsum("bhqk,bhkd->bhqd", P, v))
    ob, c_abn, f2 = batch_norm_fwd(oc, p["attention_bn/gamma"], p["attention_bn/beta"])
    y = ob @ p["output_transform/kernel"] + p["output_transform/bias"]
    return y, (x, q, k, v, P, c_lbn, c_abn, ob, h), {"logits_bn": f1, "attention_bn": f2}


def mha_bn_bwd(dy, cache, p):
    x, q, k, v, P, c_lbn, c_abn, ob, h = cache
    F = x.shape[-1]
    g = {"output_transform/kernel": ob.reshape(-1, F).T @ dy.reshape(-1, dy.shape[-1]), "output_transform/bias": dy.sum((0, 1))}
    dob = dy @ p["output_transform/kernel"].T
    doc, g["attention_bn/gamma"], g["attention_bn/beta"] = batch_norm_bwd(dob, c_abn)
    do = _split(doc, h)
    dv = np.einsum("bhqk,bhqd->bhkd", P, do)
    dP = np.einsum("bhqd,bhkd->bhqk", do, v)
    dlb = P * (dP - (P * dP).sum(-1, keepdims=True))
    dlogits, g["logits_bn/gamma"], g["logits_bn/beta"] = batch_norm_bwd(dlb, c_lbn)
    dq = _combine(np.einsum("bhqk,bhkd->bhqd", dlogits, k))
    dk = _combine(np.einsum("bhqk,bhqd->bhkd", dlogits, q))
    dv = _combine(dv)
    x2 = x.reshape(-1, F)
    g["q/kernel"], g["k/kernel"], g["v/kernel"] = x2.T @ dq.reshape(-1, F), x2.T @ dk.reshape(-1, F), x2.T @ dv.reshape(-1, F)
    dx = dq @ p["q/kernel"].T + dk @ p["k/kernel"].T + dv @ p["v/kernel"].T
    return dx, g


def _dense_relu_fwd(x, W, b):
    z = x @ W + b
    return np.maximum(z, 0.0), (x, z)


def _dense_relu_bwd(da, cache, W):
    x, z = cache
    dz = da * (z > 0)
    return dz @ W.T, x.reshape(-1, x.shape[-1]).T @ dz.reshape(-1, dz.shape[-1]), dz.reshape(-1, dz.shape[-1]).sum(0)


def transformer_encoder_fwd(x, p, h, sid):
    """TransformerEncoder.forward (transformer_utils.py:399-413) around FeedForwardNetwork.forward (:696-715):
    y = LN(MHA(x, x) + x);  n = LN(relu(relu(y W1 + b1) W2 + b2) + y)  [the FFN's own residual + layer norm];  out = LN(n + y)."""
    a, c_mha = mha_fwd(x, p, h)
    y, c_ln0 = layer_norm_fwd(a + x, p["LayerNorm/gamma"], p["LayerNorm/beta"])
    f, c_d1 = _dense_relu_fwd(y, p[f"filter_output{sid}/kernel"], p[f"filter_output{sid}/bias"])
    o, c_d2 = _dense_relu_fwd(f, p[f"ff_output{sid}/kernel"], p[f"ff_output{sid}/bias"])
    n, c_ln1 = layer_norm_fwd(o + y, p["LayerNorm_1/gamma"], p["LayerNorm_1/beta"])
    out, c_ln2 = layer_norm_fwd(n + y, p["LayerNorm_2/gamma"], p["LayerNorm_2/beta"])
    return out, (c_mha, c_ln0, c_d1, c_d2, c_ln1, c_ln2, sid)


def transformer_encoder_bwd(dout, cache, p):
    c_mha, c_ln0, c_d1, c_d2, c_ln1, c_ln2, sid = cache
    g = {}
    ds2, g["LayerNorm_2/gamma"], g["LayerNorm_2/beta"] = layer_norm_bwd(dout, c_ln2)            # d(n + y)
    ds1, g["LayerNorm_1/gamma"], g["LayerNorm_1/beta"] = layer_norm_bwd(ds2, c_ln1)             # d(o + y)
    df, g[f"ff_output{sid}/kernel"], g[f"ff_output{sid}/bias"] = _dense_relu_bwd(ds1, c_d2, p[f"ff_output{sid}/kernel"])
    dy1, g[f"filter_output{sid}/kernel"], g[f"filter_output{sid}/bias"] = _dense_relu_bwd(df, c_d1, p[f"filter_output{sid}/kernel"])
    dy = ds2 + ds1 + dy1                                                                         # y feeds three places
    ds0, g["LayerNorm/gamma"], g["LayerNorm/beta"] = layer_norm_bwd(dy, c_ln0)                   # d(a + x)
    dx, gm = mha_bwd(ds0, c_mha, p)
    g.update(gm)
    return dx + ds0, g


def transformer_encoder_mod_fwd(x, p, h, sid, keep_mask, rate=0.9):
    """TransformerEncoderMod.forward (transformer_utils.py:443-457) + FeedForwardNetworkMod.forward (:737-766), training mode:
    a = dropout(MHA_BN(x, x), rate = 1 - attention_dropout = 0.9) -- keep_mask given, kept entries scaled by 1 / (1 - rate);
    y = LN(a + x);  f = BN(relu(y W1 + b1));  out = BN(relu(f W2 + b2)) [B, L, final_size].  No residual around the FFN."""
    a, c_mha, feeds = mha_bn_fwd(x, p, h)
    a = a * keep_mask / (1.0 - rate)
    y, c_ln = layer_norm_fwd(a + x, p["LayerNorm/gamma"], p["LayerNorm/beta"])
    f0, c_d1 = _dense_relu_fwd(y, p[f"filter_output{sid}/kernel"], p[f"filter_output{sid}/bias"])
    f, c_b1, fb1 = batch_norm_fwd(f0, p["filter_bn/gamma"], p["filter_bn/beta"])
    o0, c_d2 = _dense_relu_fwd(f, p[f"ff_output{sid}/kernel"], p[f"ff_output{sid}/bias"])
    out, c_b2, fb2 = batch_norm_fwd(o0, p["feed_output_bn/gamma"], p["feed_output_bn/beta"])
    feeds.update({"filter_bn": fb1, "feed_output_bn": fb2})
    return out, (c_mha, c_ln, c_d1, c_b1, c_d2, c_b2, sid, keep_mask, rate), feeds


def transformer_encoder_mod_bwd(dout, cache, p):
    c_mha, c_ln, c_d1, c_b1, c_d2, c_b2, sid, keep_mask, rate = cache
    g = {}
    do0, g["feed_output_bn/gamma"], g["feed_output_bn/beta"] = batch_norm_bwd(dout, c_b2)
    df, g[f"ff_output{sid}/kernel"], g[f"ff_output{sid}/bias"] = _dense_relu_bwd(do0, c_d2, p[f"ff_output{sid}/kernel"])
    df0, g["filter_bn/gamma"], g["filter_bn/beta"] = batch_norm_bwd(df, c_b1)
    dy, g[f"filter_output{sid}/kernel"], g[f"filter_output{sid}/bias"] = _dense_relu_bwd(df0, c_d1, p[f"filter_output{sid}/kernel"])
    ds, g["LayerNorm/gamma"], g["LayerNorm/beta"] = layer_norm_bwd(dy, c_ln)                     # d(a + x)
    dx, gm = mha_bn_bwd(ds * keep_mask / (1.0 - rate), c_mha, p)
    g.update(gm)
    return dx + ds, g


def netvlad_atten_cluster_fwd(x2d, p, S, h, keep_mask, rate=0.9):
    """NetVladAttenCluster.forward (video_pooling_modules.py:1617-1663): similarities = TransformerEncoderMod(frames) [B, S, C] (no
    softmax: they may be negative), residual_sum[b, f, c] = sum_n sims[b, n, c] (x[b, n, f] - centres[f, c]), L2 over f per cluster,
    flatten f-major, L2 over everything."""
    F = x2d.shape[1]
    B = x2d.shape[0] // S
    enc = {k[len("cluster_attention/"):]: v for k, v in p.items() if k.startswith("cluster_attention/")}
    sims, c_enc, feeds = transformer_encoder_mod_fwd(x2d.reshape(B, S, F), enc, h, "encode", keep_mask, rate)
    f = netvlad_forward(x2d, None, None, None, p["cluster_centers"][None], B, S, residual=True, softmax_on=False, assign=sims)
    return f["out"], (x2d, sims, c_enc, enc, B, S), feeds


def netvlad_atten_cluster_bwd(dout, cache, p):
    x2d, sims, c_enc, enc, B, S = cache
    gp, _ = netvlad_backward(x2d, None, None, None, p["cluster_centers"][None], B, S, dout, residual=True, softmax_on=False, assign=sims)
    dx_enc, ge = transformer_encoder_mod_bwd(gp["assign"], c_enc, enc)
    g = {"cluster_attention/" + k: v for k, v in ge.items()}
    g["cluster_centers"] = gp["W2"].reshape(p["cluster_centers"].shape)
    return gp["x"] + dx_enc.reshape(x2d.shape), g


def combine_gradients(tower_grads):
    """utils.py:192-213: per variable, tf.stack over the towers + reduce_sum -- a SUM, not a mean."""
    return {n: np.sum(np.stack([t[n] for t in tower_grads], 0), 0) for n in tower_grads[0]}


def clip_by_norm(g, clip_norm):
    """tf.clip_by_norm as utils.py:170-189 applies it, per variable: g * clip / max(||g||_2, clip)."""
    return g * (clip_norm / max(float(np.sqrt((g * g).sum())), clip_norm))


def learning_rate(base, decay, decay_examples, global_step, batch_size, num_towers):
    """train.py:244-252: tf.train.exponential_decay(base, global_step * batch_size * num_towers, decay_examples, decay, staircase)."""
    return base * decay ** ((global_step * batch_size * num_towers) // decay_examples)
