"""Tier 2: analytic known-answer tests derivable from the cited reference lines without TF.
The reference has no tests or golden vectors for this path (PARITY UNPINNED): these pin the oracle."""
import math

import numpy as np
import torch

from oracle import lpm_oracle as O
from oracle import numpy_ref as R


def _netvlad_params(D, K, W=None, W2=None):
    return {"s/cluster_weights": torch.zeros(D, K, dtype=torch.float64) if W is None else W,
            "s/cluster_bn/gamma": torch.ones(K, dtype=torch.float64), "s/cluster_bn/beta": torch.zeros(K, dtype=torch.float64),
            "s/cluster_weights2": torch.randn(1, D, K, dtype=torch.float64) if W2 is None else W2}


def test_zero_cluster_weights_gives_uniform_assignment():
    """cluster_weights = 0, BN at init => softmax uniform => vlad = (1/K) sum_t x - (T/K) W2
    (frame_level_models.py:2781-2817)."""
    B, T, D, K = 2, 5, 6, 4
    torch.manual_seed(0)
    x = torch.randn(B * T, D, dtype=torch.float64)
    p = _netvlad_params(D, K)
    out = O.netvlad_forward(x, p, "s", T, True, True).reshape(B, D, K)
    U = x.reshape(B, T, D).sum(1)[:, :, None] / K - (T / K) * p["s/cluster_weights2"]
    N = U / U.norm(dim=1, keepdim=True)
    exp = N / N.reshape(B, -1).norm(dim=1)[:, None, None]
    assert torch.allclose(out, exp, atol=1e-12)


def test_single_cluster():
    """K = 1 => assignment == 1 => vlad = sum_t x - T*W2, L2-normalised."""
    B, T, D = 3, 4, 5
    torch.manual_seed(1)
    x = torch.randn(B * T, D, dtype=torch.float64)
    p = _netvlad_params(D, 1, W=torch.randn(D, 1, dtype=torch.float64))
    out = O.netvlad_forward(x, p, "s", T, True, True)
    U = x.reshape(B, T, D).sum(1) - T * p["s/cluster_weights2"].reshape(1, D)
    assert torch.allclose(out, U / U.norm(dim=1, keepdim=True), atol=1e-12)


def test_norm_invariants_and_frame_permutation():
    B, T, D, K = 3, 9, 8, 4
    torch.manual_seed(2)
    x = torch.randn(B, T, D, dtype=torch.float64)
    p = _netvlad_params(D, K, W=torch.randn(D, K, dtype=torch.float64))
    out = O.netvlad_forward(x.reshape(B * T, D), p, "s", T, True, True).reshape(B, D, K)
    assert torch.allclose(out.reshape(B, -1).norm(dim=1), torch.ones(B, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(out.norm(dim=1), torch.full((B, K), 1 / math.sqrt(K), dtype=torch.float64), atol=1e-12)
    perm = torch.randperm(T)
    out2 = O.netvlad_forward(x[:, perm].reshape(B * T, D), p, "s", T, True, True).reshape(B, D, K)
    assert torch.allclose(out, out2, atol=1e-12)


def test_all_zero_column_hits_the_l2_clamp():
    """tf.nn.l2_normalize floors the squared norm at 1e-12: an all-zero cluster column stays 0 (no NaN)."""
    B, T, D, K = 1, 3, 4, 3
    sims = torch.ones(B, T, K, dtype=torch.float64)
    sims[:, :, 1] = 0
    x = torch.randn(B, T, D, dtype=torch.float64)
    out = O.vlad_aggregate(sims, x, torch.randn(D, K, dtype=torch.float64)).reshape(B, D, K)
    assert torch.isfinite(out).all() and out[:, :, 1].abs().max() == 0
    assert torch.allclose(out.reshape(B, -1).norm(dim=1), torch.ones(B, dtype=torch.float64))


def test_sample_uniform_frames_index_formula():
    """idx[b,j] = floor(j*nf/S) for the sizes in scope; never reads padding (model_utils.py:112-118)."""
    for S in (30, 256, 300):
        nf = np.arange(1, 301)
        idx = O.sample_uniform_frame_index(nf, S)
        exp = (np.arange(S)[None, :] * nf[:, None]) // S
        assert (idx == exp).all()
        assert (idx.max(axis=1) < nf).all()


def test_moe_zero_weights():
    """Zero MoE weights: gate softmax = 1/(m+1), experts = 0.5 => every prediction m/(m+1)*0.5
    (video_level_models.py:116-126)."""
    for m in (2, 4):
        V, H = 7, 5
        p = {"gates/weights": torch.zeros(H, V * (m + 1)), "experts/weights": torch.zeros(H, V * m),
             "experts/biases": torch.zeros(V * m)}
        pred = O.moe_forward(torch.randn(3, H), p, V, m)
        assert torch.allclose(pred, torch.full((3, V), m / (m + 1) * 0.5))


def test_cross_entropy_at_half():
    """p = 0.5 everywhere => V * (-log(0.5 + 1e-5)) (losses.py:46-51)."""
    V = 11
    lab = torch.zeros(4, V, dtype=torch.bool)
    lab[:, ::3] = True
    loss = O.cross_entropy_loss(torch.full((4, V), 0.5, dtype=torch.float64), lab)
    assert abs(float(loss) - V * (-math.log(0.5 + 1e-5))) < 1e-12
    assert abs(R.cross_entropy(np.full((4, V), 0.5), lab.numpy()) - float(loss)) < 1e-12


def test_layer_norm_is_joint_over_length_and_features():
    """tf.contrib.layers.layer_norm default begin_norm_axis=1: one mean/variance per example."""
    x = torch.randn(2, 5, 7, dtype=torch.float64)
    p = {"ln/gamma": torch.ones(7, dtype=torch.float64), "ln/beta": torch.zeros(7, dtype=torch.float64)}
    y = O.layer_norm(x, p, "ln")
    assert torch.allclose(y.reshape(2, -1).mean(1), torch.zeros(2, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(y.reshape(2, -1).var(1, unbiased=False), torch.ones(2, dtype=torch.float64), atol=1e-9)


def test_adam_is_tf_flavoured():
    """epsilon is added to the un-bias-corrected sqrt(v) (differs from torch.optim.Adam)."""
    p, g = torch.tensor([1.0], dtype=torch.float64), torch.tensor([1e-9], dtype=torch.float64)
    new, m, v = O.adam_tf_update(p, g, torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64), 0.1, 1)
    lr_t = 0.1 * math.sqrt(1 - 0.999) / (1 - 0.9)
    exp = 1.0 - lr_t * (0.1 * 1e-9) / (math.sqrt(0.001 * 1e-18) + 1e-8)
    assert abs(float(new) - exp) < 1e-15
    n2, _, _ = R.adam_tf(np.array([1.0]), np.array([1e-9]), np.zeros(1), np.zeros(1), 0.1, 1)
    assert abs(float(n2[0]) - exp) < 1e-15


def test_clip_and_combine_semantics():
    """SUM over towers (utils.py:207-211), then per-variable clip t*c/max(||t||,c) (utils.py:181-188)."""
    a = {"w": torch.tensor([3.0, 4.0]), "b": torch.tensor([0.1])}
    b = {"w": torch.tensor([3.0, 4.0]), "b": torch.tensor([0.1])}
    s = O.combine_gradients([a, b])
    assert torch.allclose(s["w"], torch.tensor([6.0, 8.0]))
    c = O.clip_gradient_norms(s, 1.0)
    assert torch.allclose(c["w"], torch.tensor([0.6, 0.8])) and torch.allclose(c["b"], torch.tensor([0.2]))


def test_learning_rate_staircase():
    cfg = O.OracleConfig(base_learning_rate=2e-4, learning_rate_decay=0.85, learning_rate_decay_examples=4000000)
    assert O.learning_rate(cfg, 0, 80, 8) == 2e-4
    assert O.learning_rate(cfg, 6249, 80, 8) == 2e-4                # 3 999 360 examples
    assert abs(O.learning_rate(cfg, 6250, 80, 8) - 2e-4 * 0.85) < 1e-18


def test_two_tower_step_equals_sum_of_tower_gradients():
    """train.py:266-336: the global batch is split, tower gradients are SUMMED (not averaged)."""
    cfg = O.OracleConfig(model="NetVladV1", iterations=6, cluster_size=8, hidden_size=16, vocab_size=20)
    x, nf, lab = O.make_synthetic_batch(4, 8, 1152, 20, seed=3, min_frames=4)
    p = {k: v.double() for k, v in O.init_params(cfg, 1152, seed=5).items()}
    _, _, info = O.train_step(p, {"step": 0, "m": {}, "v": {}}, x.double(), nf, lab, cfg, 2)
    g0 = O.loss_and_grads(p, x[:2].double(), nf[:2], lab[:2], cfg)[2]
    g1 = O.loss_and_grads(p, x[2:].double(), nf[2:], lab[2:], cfg)[2]
    ref = O.clip_gradient_norms(O.combine_gradients([g0, g1]), 1.0)
    for n in ref:
        assert torch.allclose(info["clipped_grads"][n], ref[n], atol=1e-14)
