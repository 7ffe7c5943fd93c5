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


# ---- round 3: the numpy restatement beyond NetVLAD (hand-derived backward) against lpm_oracle.py (torch autograd), fp64 ------------
def _t(a, grad=False):
    return torch.tensor(a, dtype=torch.float64, requires_grad=grad)


def _close(a, b, what, rtol=1e-8, atol=1e-10):
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, b, rtol=rtol, atol=atol, err_msg=what)


def test_layer_norm_is_joint_over_all_non_batch_axes():
    """tf.contrib.layers.layer_norm defaults: ONE mean / variance per example over (L, F) -- not per token (App. B)."""
    rng = np.random.default_rng(11)
    x, gamma, beta, dy = rng.standard_normal((3, 5, 8)), 1 + 0.3 * rng.standard_normal(8), 0.2 * rng.standard_normal(8), rng.standard_normal((3, 5, 8))
    y, cache = R.layer_norm_fwd(x, gamma, beta)
    xt, p = _t(x, True), {"s/gamma": _t(gamma, True), "s/beta": _t(beta, True)}
    yt = O.layer_norm(xt, p, "s")
    _close(yt, y, "layer_norm forward")
    # per-example statistics of the normalised tensor: mean 0, variance 1 over all 40 entries jointly
    z = (y - beta) / gamma
    np.testing.assert_allclose(z.mean((1, 2)), 0, atol=1e-12)
    np.testing.assert_allclose((z ** 2).mean((1, 2)), 1, rtol=1e-9)
    yt.backward(_t(dy))
    dx, dg, db = R.layer_norm_bwd(dy, cache)
    _close(xt.grad, dx, "layer_norm dx")
    _close(p["s/gamma"].grad, dg, "layer_norm dgamma")
    _close(p["s/beta"].grad, db, "layer_norm dbeta")


def test_batch_norm_rank_2_3_4_and_moving_variance_feed():
    """slim.batch_norm: channel = last axis at every rank; the moving variance is fed the UNBIASED estimate on the fused (rank 2 / 4)
    path and the biased one at rank 3 (App. B)."""
    rng = np.random.default_rng(12)
    for shape in ((9, 6), (3, 5, 6), (2, 3, 4, 6)):
        x, gamma, beta, dy = rng.standard_normal(shape), 1 + 0.3 * rng.standard_normal(6), 0.2 * rng.standard_normal(6), rng.standard_normal(shape)
        y, cache, fed = R.batch_norm_fwd(x, gamma, beta)
        xt, p = _t(x, True), {"s/gamma": _t(gamma, True), "s/beta": _t(beta, True)}
        upd = {}
        yt = O.batch_norm(xt, p, "s", True, upd)
        _close(yt, y, f"batch_norm forward {shape}")
        _close(upd["s/moving_mean"], fed["moving_mean"], "moving mean feed")
        _close(upd["s/moving_variance"], fed["moving_variance"], "moving variance feed")
        n = x.size // 6
        biased = x.reshape(-1, 6).var(0)
        want = biased * n / (n - 1) if len(shape) in (2, 4) else biased
        np.testing.assert_allclose(fed["moving_variance"], want, rtol=1e-12)
        yt.backward(_t(dy))
        dx, dg, db = R.batch_norm_bwd(dy, cache)
        _close(xt.grad, dx, f"batch_norm dx {shape}")
        _close(p["s/gamma"].grad, dg, "batch_norm dgamma")
        _close(p["s/beta"].grad, db, "batch_norm dbeta")


def _encoder_params(rng, F, ff, sid, bn=None, final=None, L=None):
    final = F if final is None else final
    p = {"q/kernel": rng.standard_normal((F, F)) / F ** 0.5, "k/kernel": rng.standard_normal((F, F)) / F ** 0.5,
         "v/kernel": rng.standard_normal((F, F)) / F ** 0.5, "output_transform/kernel": rng.standard_normal((F, F)) / F ** 0.5,
         "output_transform/bias": 0.1 * rng.standard_normal(F),
         "LayerNorm/gamma": 1 + 0.2 * rng.standard_normal(F), "LayerNorm/beta": 0.1 * rng.standard_normal(F),
         f"filter_output{sid}/kernel": rng.standard_normal((F, ff)) / F ** 0.5, f"filter_output{sid}/bias": 0.3 * rng.standard_normal(ff),
         f"ff_output{sid}/kernel": rng.standard_normal((ff, final)) / ff ** 0.5, f"ff_output{sid}/bias": 0.3 * rng.standard_normal(final)}
    if bn:
        for name, c in (("logits_bn", L), ("attention_bn", F), ("filter_bn", ff), ("feed_output_bn", final)):
            p[name + "/gamma"], p[name + "/beta"] = 1 + 0.2 * rng.standard_normal(c), 0.1 * rng.standard_normal(c)
    else:
        for name in ("LayerNorm_1", "LayerNorm_2"):
            p[name + "/gamma"], p[name + "/beta"] = 1 + 0.2 * rng.standard_normal(F), 0.1 * rng.standard_normal(F)
    return p


def test_transformer_encoder_v1_forward_and_every_gradient():
    """TransformerEncoder (transformer_utils.py:399-413, 696-715): three layer norms, ReLU on BOTH dense layers of the FFN, the FFN's
    input feeding three places -- numpy chain rule vs torch autograd."""
    rng = np.random.default_rng(13)
    B, L, F, h, ff = 3, 6, 16, 4, 24
    p = _encoder_params(rng, F, ff, "encode1")
    x, dout = rng.standard_normal((B, L, F)), rng.standard_normal((B, L, F))
    out, cache = R.transformer_encoder_fwd(x, p, h, "encode1")
    pt = {"enc/" + k: _t(v, True) for k, v in p.items()}
    xt = _t(x, True)
    ot = O.transformer_encoder(xt, pt, "enc", h, "encode1")
    _close(ot, out, "V1 encoder forward")
    ot.backward(_t(dout))
    dx, g = R.transformer_encoder_bwd(dout, cache, p)
    _close(xt.grad, dx, "V1 encoder dx")
    assert set(g) == set(p)
    for k in p:
        _close(pt["enc/" + k].grad, g[k], f"V1 encoder d {k}")


def test_transformer_encoder_mod_v2_forward_and_every_gradient():
    """TransformerEncoderMod (transformer_utils.py:443-457, 634-677, 737-766): batch norm over the KEY-POSITION channel of the rank-4
    logits, no q scaling, dropout rate 0.9 through a given keep mask, two batch norms in the FFN, no residual around it."""
    rng = np.random.default_rng(14)
    B, L, F, h, ff, final = 3, 7, 16, 2, 24, 5
    p = _encoder_params(rng, F, ff, "encode", bn=True, final=final, L=L)
    x, dout = rng.standard_normal((B, L, F)), rng.standard_normal((B, L, final))
    keep = (rng.random((B, L, F)) >= 0.5).astype(np.float64)
    out, cache, feeds = R.transformer_encoder_mod_fwd(x, p, h, "encode", keep, rate=0.9)
    pt = {"enc/" + k: _t(v, True) for k, v in p.items()}
    xt = _t(x, True)
    upd = {}
    ot = O.transformer_encoder_mod(xt, pt, "enc", h, "encode", True, 0.9, _t(keep), upd)
    _close(ot, out, "V2 encoder forward")
    for scope, f in feeds.items():
        _close(upd[f"enc/{scope}/moving_mean"], f["moving_mean"], scope + " moving mean feed")
        _close(upd[f"enc/{scope}/moving_variance"], f["moving_variance"], scope + " moving variance feed")
    assert feeds["logits_bn"]["moving_mean"].shape == (L,), "logits_bn's channel is the key position"
    ot.backward(_t(dout))
    dx, g = R.transformer_encoder_mod_bwd(dout, cache, p)
    _close(xt.grad, dx, "V2 encoder dx")
    assert set(g) == set(p)
    for k in p:
        _close(pt["enc/" + k].grad, g[k], f"V2 encoder d {k}", rtol=1e-7, atol=1e-9)


def test_netvlad_atten_cluster_forward_and_every_gradient():
    """NetVladAttenCluster (video_pooling_modules.py:1617-1663): encoder similarities (not a distribution) -> residual sums against
    cluster_centers -> L2 per cluster over F -> f-major flatten -> L2; also against the reference's literal [B, N, F, C] formulation."""
    rng = np.random.default_rng(15)
    B, S, F, K = 2, 5, 32, 6                    # heads = F // 16 = 2 (video_pooling_modules.py:1613)
    enc = _encoder_params(rng, F, 4 * F, "encode", bn=True, final=K, L=S)
    p = {"cluster_attention/" + k: v for k, v in enc.items()}
    p["cluster_centers"] = rng.standard_normal((F, K)) / F ** 0.5
    x, dout = rng.standard_normal((B * S, F)), rng.standard_normal((B, F * K))
    keep = (rng.random((B, S, F)) >= 0.4).astype(np.float64)
    out, cache, _ = R.netvlad_atten_cluster_fwd(x, p, S, F // 16, keep, rate=0.9)
    pt = {"s/" + k: _t(v, True) for k, v in p.items()}
    xt = _t(x, True)
    ot = O.netvlad_atten_cluster_forward(xt, pt, "s", S, True, 0.9, _t(keep))
    _close(ot, out, "V2 aggregator forward")
    with torch.no_grad():
        literal = O.netvlad_atten_cluster_forward(_t(x), pt, "s", S, True, 0.9, _t(keep), explicit_4d=True)
    _close(literal, out, "V2 aggregator forward, the reference's 4-d formulation")
    ot.backward(_t(dout))
    dx, g = R.netvlad_atten_cluster_bwd(dout, cache, p)
    _close(xt.grad, dx, "V2 aggregator dx", rtol=1e-7, atol=1e-9)
    assert set(g) == set(p)
    for k in p:
        _close(pt["s/" + k].grad, g[k], f"V2 aggregator d {k}", rtol=1e-7, atol=1e-9)


def test_combine_clip_and_learning_rate():
    """utils.py:170-213 (SUM over towers, per-variable clip_by_norm) and train.py:244-252 (staircase decay on examples seen)."""
    rng = np.random.default_rng(16)
    towers = [{"a": rng.standard_normal((4, 3)), "b": 5 * rng.standard_normal(7)} for _ in range(3)]
    s = R.combine_gradients(towers)
    st = O.combine_gradients([{k: _t(v) for k, v in t.items()} for t in towers])
    ct = O.clip_gradient_norms(st, 1.0)
    for k in s:
        _close(st[k], s[k], "combine " + k)
        np.testing.assert_allclose(s[k], towers[0][k] + towers[1][k] + towers[2][k], rtol=1e-12)
        c = R.clip_by_norm(s[k], 1.0)
        _close(ct[k], c, "clip " + k)
        assert np.linalg.norm(c) <= 1.0 + 1e-12
    small = 0.01 * towers[0]["a"]
    np.testing.assert_array_equal(R.clip_by_norm(small, 1.0), small)          # below the bound: untouched
    cfg = O.OracleConfig(base_learning_rate=2e-4, learning_rate_decay=0.85, learning_rate_decay_examples=4000000)
    for step, bs, n in ((0, 80, 1), (49999, 80, 1), (50000, 80, 1), (6250, 80, 8), (31250, 80, 8)):
        assert R.learning_rate(2e-4, 0.85, 4000000, step, bs, n) == O.learning_rate(cfg, step, bs, n)
    assert R.learning_rate(2e-4, 0.85, 4000000, 6250, 80, 8) == 2e-4 * 0.85


def test_logits_bn_backward_correction_is_affine_in_the_score():
    """The identity behind lpm_mha_bn_dk_correct: with ds = sck dz - ca - s cb and s = q . k, the share of
    the two correction terms in dK[j] = sum_q ds[q, j] q is  - ca[j] Sq - cb[j] Qm k[j]  (Sq = sum_q q, Qm = sum_q q q^T), and in
    dQ[q] = sum_j ds[q, j] k[j] it is  - sum_j ca[j] k[j] - (sum_j cb[j] k[j] k[j]^T) q.  fp64, one (batch, head)."""
    import numpy as np
    rng = np.random.default_rng(5)
    L, d = 37, 16
    q, k = rng.normal(size=(L, d)), rng.normal(size=(L, d))
    dz, sck, ca, cb = rng.normal(size=(L, L)), rng.uniform(0.5, 1.5, L), rng.normal(size=L) * 0.1, rng.normal(size=L) * 0.1
    s = q @ k.T
    ds = dz * sck[None, :] - ca[None, :] - s * cb[None, :]
    dk, dq = ds.T @ q, ds @ k
    ds0 = dz * sck[None, :]
    Sq, Qm = q.sum(0), q.T @ q
    dk_fix = (ds0.T @ q) - ca[:, None] * Sq[None, :] - cb[:, None] * (k @ Qm)
    assert np.allclose(dk_fix, dk, rtol=1e-12, atol=1e-12)
    Mk = (k * cb[:, None]).T @ k
    dq_fix = (ds0 @ k) - (ca[:, None] * k).sum(0)[None, :] - q @ Mk
    assert np.allclose(dq_fix, dq, rtol=1e-12, atol=1e-12)
