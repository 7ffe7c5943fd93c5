"""Torch-CPU restatement (dtype-generic: fp64 for pinning, fp32 for timing) of the
reference's NetVladV1 / NetVladV2 training step.  TEST INFRASTRUCTURE ONLY and
PARITY UNPINNED -- see ``oracle/__init__.py``.

Reference = /root/reference (pomonam/LearnablePoolingMethods).  Every function
cites the reference lines it restates.  Weights live in a flat ``dict`` keyed
by the TF1 variable names of SURVEY.md App. A.9 (without the ``tower/`` prefix)
so the product's variable store and the oracle exchange weights by name.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

__all__ = [
    "OracleConfig", "l2_normalize", "batch_norm", "layer_norm", "sample_uniform_frame_index",
    "sample_uniform_frames", "netvlad_forward", "lightvlad_forward", "netvlad_atten_cluster_forward",
    "vlad_aggregate", "multi_head_attention", "multi_head_attention_bn", "attention_core",
    "transformer_encoder", "transformer_encoder_mod", "one_fc_attention_forward", "attention_modules_mha",
    "transformer_encoder_block", "moe_forward", "cross_entropy_loss",
    "model_forward", "init_params", "trainable_names", "loss_and_grads", "combine_gradients",
    "clip_gradient_norms", "learning_rate", "adam_tf_update", "train_step", "make_synthetic_batch",
    "BN_EPS", "BN_DECAY", "LN_EPS", "L2N_EPS",
]

BN_EPS = 1e-3      # slim.batch_norm default epsilon
BN_DECAY = 0.999   # slim.batch_norm default decay
LN_EPS = 1e-12     # tf.contrib.layers.layer_norm variance epsilon
L2N_EPS = 1e-12    # tf.nn.l2_normalize epsilon


@dataclass
class OracleConfig:
    """Flag defaults: frame_level_models.py:35,2197-2207; video_level_models.py:26-36;
    train.py:78-108; README.md:12-18 for the values the metric uses."""
    model: str = "NetVladV1"            # "NetVladV1" | "NetVladV2" | "WillowModelReg" 
    iterations: int = 30                # frame_level_models.py:35
    cluster_size: int = 256             # :2199
    hidden_size: int = 1024             # :2201
    add_batch_norm: bool = True         # :2197
    relu: bool = False                  # :2203 netvlad_relu
    gating: bool = True                 # :2205
    remove_diag: bool = False           # :2207
    encoder: bool = True                # False = "gated NetVLAD" of BASELINE cfg-5 (no cluster encoders)
    moe_num_mixtures: int = 2           # video_level_models.py:27
    moe_l2: float = 1e-8                # :35
    moe_low_rank_gating: int = -1       # :37-39 (-1: one gates layer)
    moe_prob_gating: bool = False       # :40-42
    moe_prob_gating_input: str = "prob"  # :43-45
    vocab_size: int = 3862              # readers.py:144
    v2_dropout_rate: float = 0.9        # transformer_utils.py:450 (rate = 1 - 0.1), App. C10
    # training (train.py:78-108)
    base_learning_rate: float = 0.01
    learning_rate_decay: float = 0.95
    learning_rate_decay_examples: float = 4000000.0
    regularization_penalty: float = 1.0
    clip_gradient_norm: float = 1.0
    rgb_det_reg: float = 1e-4           # frame_level_models.py:2213 (WillowModelReg / NetVladOrthoReg)
    audio_det_reg: float = 1e-4         # :2209
    sample_random_frames: bool = True   # :40
    video_dim: int = 1024               # frame_level_models.py:2261,2274
    audio_dim: int = 128                # :2263,2277


# --------------------------------------------------------------------------------------
# TF1 op semantics
# --------------------------------------------------------------------------------------
def l2_normalize(x: torch.Tensor, dim: int, eps: float = L2N_EPS) -> torch.Tensor:
    """tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), eps)) (train.py:264; frame_level_models.py:2819,2822)."""
    ss = (x * x).sum(dim=dim, keepdim=True)
    return x * torch.rsqrt(torch.clamp(ss, min=eps))


def batch_norm(x, params, scope, is_training, updates=None):
    """slim.batch_norm(center, scale): channel = last axis, stats over all others; training
    normalises with the biased batch variance (eps 1e-3); the moving variance is fed the
    unbiased variance on the fused (rank-2/4) path (frame_level_models.py:2266,2784,2355;
    transformer_utils.py:653,666,747,760)."""
    gamma, beta = params[scope + "/gamma"], params[scope + "/beta"]
    if is_training:
        red = tuple(range(x.dim() - 1))
        n = x.numel() // x.shape[-1]
        mean = x.mean(dim=red)
        var = ((x - mean) ** 2).mean(dim=red)
        if updates is not None:
            uvar = var * (n / max(n - 1, 1)) if x.dim() in (2, 4) else var
            updates[scope + "/moving_mean"] = mean.detach()
            updates[scope + "/moving_variance"] = uvar.detach()
    else:
        mean, var = params[scope + "/moving_mean"], params[scope + "/moving_variance"]
    return (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta


def layer_norm(x, params, scope):
    """tf.contrib.layers.layer_norm defaults: begin_norm_axis=1 => moments over every
    non-batch axis jointly, gamma/beta on the last axis, eps 1e-12
    (transformer_utils.py:407,411,454,713)."""
    red = tuple(range(1, x.dim()))
    mean = x.mean(dim=red, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=red, keepdim=True)
    return (x - mean) * torch.rsqrt(var + LN_EPS) * params[scope + "/gamma"] + params[scope + "/beta"]


# --------------------------------------------------------------------------------------
# a2: SampleUniformFrames (model_utils.py:101-122)
# --------------------------------------------------------------------------------------
def sample_uniform_frame_index(num_frames, num_samples: int) -> np.ndarray:
    """idx[b, j] = int32(fp32(linspace(0,1,S+1)[j]) * fp32(nf[b])) (model_utils.py:113-118).
    tf.linspace evaluates start + j*step in fp32; the cast truncates toward zero."""
    nf = np.asarray(num_frames, dtype=np.float32).reshape(-1, 1)
    step = np.float32(1.0) / np.float32(num_samples)
    v = (np.arange(num_samples, dtype=np.float32) * step).astype(np.float32)
    v = np.minimum(v, np.float32(1.0))
    return (v[None, :] * nf).astype(np.float32).astype(np.int32)


def sample_uniform_frames(model_input: torch.Tensor, num_frames, num_samples: int) -> torch.Tensor:
    """tf.gather_nd over (batch, frame) pairs (model_utils.py:119-122)."""
    nf = num_frames.detach().cpu().numpy() if torch.is_tensor(num_frames) else num_frames
    idx = torch.from_numpy(sample_uniform_frame_index(nf, num_samples)).long()
    b = torch.arange(model_input.shape[0]).unsqueeze(1).expand_as(idx)
    return model_input[b, idx]


def sample_random_frames(model_input: torch.Tensor, num_frames, num_samples: int, uniform: torch.Tensor) -> torch.Tensor:
    """SampleRandomFrames (model_utils.py:60-78): idx[b,j] = int32(u[b,j] * fp32(nf[b])), u ~ U[0,1) [B, S] (given here)."""
    nf = torch.as_tensor(num_frames).reshape(-1, 1).to(torch.float32)
    idx = (uniform.to(torch.float32) * nf).to(torch.int32).long()
    b = torch.arange(model_input.shape[0]).unsqueeze(1).expand_as(idx)
    return model_input[b, idx]


def sample_random_sequence(model_input: torch.Tensor, num_frames, num_samples: int, uniform: torch.Tensor) -> torch.Tensor:
    """SampleRandomSequence (model_utils.py:26-57): a contiguous run starting at int32(u[b] * fp32(max(nf - S, 0) + 1)),
    clamped to nf - 1; u ~ U[0,1) [B, 1]."""
    nf = torch.as_tensor(num_frames).reshape(-1, 1).to(torch.float32)
    max_start = torch.clamp(nf - num_samples, min=0.0)
    start = (uniform.to(torch.float32).reshape(-1, 1) * (max_start + 1.0)).to(torch.int32)
    idx = torch.minimum(start + torch.arange(num_samples, dtype=torch.int32).unsqueeze(0), (nf - 1).to(torch.int32)).long()
    b = torch.arange(model_input.shape[0]).unsqueeze(1).expand_as(idx)
    return model_input[b, idx]


def orthogonal_regularizer(weights: torch.Tensor, scale: float) -> torch.Tensor:
    """module_utils.py:55-90: scale * sum |W^T W - I| with W = l2_normalize(weights, axis=1) (each ROW of the [D,K] matrix
    normalised over the clusters), W^T W [K,K]."""
    w = l2_normalize(weights, 1)
    det = w.transpose(0, 1) @ w - torch.eye(w.shape[1], dtype=w.dtype)
    return scale * det.abs().sum()


def netvlad_orthoreg_forward(x2d, params, scope, scope_id, S, add_batch_norm=True, is_training=True, updates=None):
    """NetVladOrthoReg.forward (video_pooling_modules.py:1520-1586): NetVLAD with a 2-D cluster_weights2 and variable names
    suffixed by the scope id ("cluster_weights<scope_id>"); the orthogonality penalty is a regulariser on cluster_weights2."""
    act = x2d @ params[f"{scope}/cluster_weights{scope_id}"]
    if add_batch_norm:
        act = batch_norm(act, params, scope + "/cluster_bn", is_training, updates)
    else:
        act = act + params[f"{scope}/cluster_biases{scope_id}"]
    assign = torch.softmax(act, dim=-1).reshape(-1, S, act.shape[-1])
    return vlad_aggregate(assign, x2d.reshape(-1, S, x2d.shape[-1]), params[scope + "/cluster_weights2"])


# --------------------------------------------------------------------------------------
# a4 + a5: NetVLAD.forward (frame_level_models.py:2773-2824), LightVLAD (:2835-2877)
# --------------------------------------------------------------------------------------
def vlad_aggregate(assign: torch.Tensor, x: torch.Tensor, centres: Optional[torch.Tensor]) -> torch.Tensor:
    """V[b,d,k] = sum_t A[b,t,k] * (x[b,t,d] - W2[d,k]) -> l2n over D -> flatten d-major -> l2n.
    frame_level_models.py:2803-2822 (and video_pooling_modules.py:1646-1658 with A := sims)."""
    B = x.shape[0]
    vlad = torch.matmul(assign.transpose(1, 2), x)           # [B,K,D]     :2812-2815
    vlad = vlad.transpose(1, 2)                              # [B,D,K]     :2816
    if centres is not None:
        a_sum = assign.sum(dim=1, keepdim=True)              # [B,1,K]     :2803
        vlad = vlad - a_sum * centres.reshape(1, centres.shape[-2], centres.shape[-1])  # :2810,2817
    vlad = l2_normalize(vlad, 1)                             # over D      :2819
    vlad = vlad.reshape(B, -1)                               # d*K + k     :2821
    return l2_normalize(vlad, 1)                             #             :2822


def _assignment(x2d, params, scope, S, add_batch_norm, is_training, updates):
    act = x2d @ params[scope + "/cluster_weights"]                                  # :2781
    if add_batch_norm:
        act = batch_norm(act, params, scope + "/cluster_bn", is_training, updates)  # :2783-2789
    else:
        act = act + params[scope + "/cluster_biases"]                               # :2790-2796
    act = torch.softmax(act, dim=-1)                                                # :2798
    return act.reshape(-1, S, act.shape[-1])                                        # :2801


def netvlad_forward(x2d, params, scope, S, add_batch_norm=True, is_training=True, updates=None):
    """NetVLAD.forward: x2d [(B*S), D] -> [B, D*K] d-major, L2-normalised."""
    assign = _assignment(x2d, params, scope, S, add_batch_norm, is_training, updates)
    x = x2d.reshape(-1, S, x2d.shape[-1])
    return vlad_aggregate(assign, x, params[scope + "/cluster_weights2"])


def lightvlad_forward(x2d, params, scope, S, add_batch_norm=True, is_training=True, updates=None):
    """LightVLAD.forward: NetVLAD without the centre-residual term (frame_level_models.py:2835-2877)."""
    assign = _assignment(x2d, params, scope, S, add_batch_norm, is_training, updates)
    return vlad_aggregate(assign, x2d.reshape(-1, S, x2d.shape[-1]), None)


# --------------------------------------------------------------------------------------
# a6 / a8: transformer blocks (transformer_utils.py)
# --------------------------------------------------------------------------------------
def _split_heads(x, h):                      # transformer_utils.py:521-540
    B, L, F = x.shape
    return x.reshape(B, L, h, F // h).permute(0, 2, 1, 3)


def _combine_heads(x):                       # :542-550
    B, h, L, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, h * d)


def attention_core(q, k, v, scale: float, logits_bn=None):
    """softmax(scale * q k^T [-> BN over key-position channel]) v on [B,h,L,d] tensors
    (transformer_utils.py:570-578; BN variant :652-661)."""
    logits = torch.matmul(q * scale, k.transpose(-1, -2))
    if logits_bn is not None:
        logits = logits_bn(logits)
    return torch.matmul(torch.softmax(logits, dim=-1), v)


def multi_head_attention(x, params, scope, num_heads):
    """MultiHeadAttention.forward(x, x): transformer_utils.py:552-586."""
    F = x.shape[-1]
    q = _split_heads(x @ params[scope + "/q/kernel"], num_heads)        # :559
    k = _split_heads(x @ params[scope + "/k/kernel"], num_heads)        # :560
    v = _split_heads(x @ params[scope + "/v/kernel"], num_heads)        # :561
    depth = F // num_heads
    o = attention_core(q, k, v, depth ** -0.5)                          # :570-578
    o = _combine_heads(o)                                               # :581
    return o @ params[scope + "/output_transform/kernel"] + params[scope + "/output_transform/bias"]  # :583


def multi_head_attention_bn(x, params, scope, num_heads, is_training, updates=None):
    """MultiHeadAttentionBN.forward(x, x): no q scaling, batch_norm on the rank-4 logits
    (channel = key position), batch_norm on the combined heads (transformer_utils.py:634-677)."""
    q = _split_heads(x @ params[scope + "/q/kernel"], num_heads)
    k = _split_heads(x @ params[scope + "/k/kernel"], num_heads)
    v = _split_heads(x @ params[scope + "/v/kernel"], num_heads)
    o = attention_core(q, k, v, 1.0,
                       lambda lg: batch_norm(lg, params, scope + "/logits_bn", is_training, updates))  # :652-659
    o = _combine_heads(o)
    o = batch_norm(o, params, scope + "/attention_bn", is_training, updates)                       # :666-671
    return o @ params[scope + "/output_transform/kernel"] + params[scope + "/output_transform/bias"]


# Test hook: when a dict, every ReLU site of the encoders records its pre-activation (detached) under the name of the bias
# variable that feeds it -- tests use it to keep their seeded weights away from pre-activations within rounding of zero, where
# the derivative of ReLU is discontinuous and ANY two correct implementations may disagree (oracle/test_weights.separate_relu_units).
RELU_TAPS: Optional[Dict[str, torch.Tensor]] = None


def _relu(z, bias_name):
    if RELU_TAPS is not None:
        RELU_TAPS[bias_name] = z.detach()
    return torch.relu(z)


def transformer_encoder(x, params, scope, num_heads, scope_id):
    """TransformerEncoder.forward (transformer_utils.py:399-413) with FeedForwardNetwork (:696-715):
    relu on both dense layers, residual+LN inside the FFN and again outside (App. C11)."""
    att = multi_head_attention(x, params, scope, num_heads) + x            # :403-405
    y = layer_norm(att, params, scope + "/LayerNorm")                      # :407
    f = _relu(y @ params[scope + f"/filter_output{scope_id}/kernel"] + params[scope + f"/filter_output{scope_id}/bias"],
              scope + f"/filter_output{scope_id}/bias")
    g = _relu(f @ params[scope + f"/ff_output{scope_id}/kernel"] + params[scope + f"/ff_output{scope_id}/bias"],
              scope + f"/ff_output{scope_id}/bias")
    n = layer_norm(g + y, params, scope + "/LayerNorm_1")                  # :712-713
    return layer_norm(n + y, params, scope + "/LayerNorm_2")               # :410-411


def transformer_encoder_mod(x, params, scope, num_heads, scope_id, is_training, dropout_rate,
                            dropout_mask=None, updates=None):
    """TransformerEncoderMod.forward (transformer_utils.py:443-457) + FeedForwardNetworkMod (:737-766)."""
    att = multi_head_attention_bn(x, params, scope, num_heads, is_training, updates)
    if is_training and dropout_rate > 0.0:                                 # :450  tf.layers.dropout
        if dropout_mask is None:
            dropout_mask = (torch.rand_like(att) >= dropout_rate).to(att.dtype)
        att = att * dropout_mask / (1.0 - dropout_rate)
    y = layer_norm(att + x, params, scope + "/LayerNorm")                  # :451-454
    f = _relu(y @ params[scope + f"/filter_output{scope_id}/kernel"] + params[scope + f"/filter_output{scope_id}/bias"],
              scope + f"/filter_output{scope_id}/bias")
    f = batch_norm(f, params, scope + "/filter_bn", is_training, updates)  # :747-752
    o = _relu(f @ params[scope + f"/ff_output{scope_id}/kernel"] + params[scope + f"/ff_output{scope_id}/bias"],
              scope + f"/ff_output{scope_id}/bias")
    return batch_norm(o, params, scope + "/feed_output_bn", is_training, updates)  # :760-765


# --------------------------------------------------------------------------------------
# a7: NetVladAttenCluster.forward (video_pooling_modules.py:1617-1663)
# --------------------------------------------------------------------------------------
def netvlad_atten_cluster_forward(x2d, params, scope, S, is_training=True, dropout_rate=0.9,
                                  dropout_mask=None, updates=None, explicit_4d=False):
    F = x2d.shape[-1]
    x = x2d.reshape(-1, S, F)                                              # :1623
    sims = transformer_encoder_mod(x, params, scope + "/cluster_attention", F // 16, "encode",
                                   is_training, dropout_rate, dropout_mask, updates)   # :1628-1638
    centres = params[scope + "/cluster_centers"]                           # [F,K] :1641-1643
    if explicit_4d:  # as written, with App. C6's intended broadcast [B,N,1,C]
        resid = x.unsqueeze(3) - centres                                   # [B,N,F,C] :1646-1647
        rsum = (resid * sims.unsqueeze(2)).sum(dim=1)                      # [B,F,C]   :1650-1652
        v = l2_normalize(rsum, 1).reshape(x.shape[0], -1)                  # :1655-1656
        return l2_normalize(v, 1)                                          # :1657
    return vlad_aggregate(sims, x, centres)


# --------------------------------------------------------------------------------------
# attention_modules.py (OneFcAttention, MultiHeadAttention, TransformerEncoderBlock)
# --------------------------------------------------------------------------------------
def one_fc_attention_forward(x2d, params, scope, num_frames, do_shift=True):
    """attention_modules.OneFcAttention.forward (attention_modules.py:29-64): one-layer attention whose softmax runs over
    the FRAMES of a clip (dim=1, :37), weighted frame sums per cluster, then shift, per-cluster L2 and 1/sqrt(K)."""
    pre = scope + "/" if scope else ""
    W = params[pre + "one_fc_attention_weight"]                            # [F,K] :30-33
    F, K = W.shape
    att = (x2d @ W).reshape(-1, num_frames, K) * (1.0 / math.sqrt(F))      # :34-36
    att = torch.softmax(att, dim=1)                                        # :37
    act = att.transpose(1, 2) @ x2d.reshape(-1, num_frames, F)             # [B,K,F] :39-41
    act = act.reshape(-1, F)                                               # :44
    if do_shift:
        act = params[pre + "alpha"] * act + params[pre + "beta"]           # :46-57
        act = l2_normalize(act, 1) * (1.0 / math.sqrt(K))                  # :58-59
    return act.reshape(-1, K * F)                                          # :61


def attention_modules_mha(x2d, params, scope, num_heads, num_units, max_frames, block_id):
    """attention_modules.MultiHeadAttention.forward (:78-112): per head its own relu(dense) q, k, v (default layer names
    dense, dense_1, dense_2 under Block<b>Layer<i>), logits divided by sqrt(num_units) TWICE (:95-96), heads concatenated."""
    pre = scope + "/" if scope else ""
    outs = []
    for i in range(num_heads):
        sc = f"{pre}Block{block_id}Layer{i}"
        q, k, v = (torch.relu(x2d @ params[f"{sc}/{n}/kernel"] + params[f"{sc}/{n}/bias"]).reshape(-1, max_frames, num_units)
                   for n in ("dense", "dense_1", "dense_2"))                # :81-89
        logits = q @ k.transpose(1, 2) / math.sqrt(num_units) / math.sqrt(num_units)   # :92-96
        outs.append(torch.softmax(logits, dim=-1) @ v)                     # :96-98
    return torch.cat(outs, dim=2)                                          # :104-110


def transformer_encoder_block(x2d, params, scope, num_units, max_frames, feature_size, num_heads, block_id):
    """attention_modules.TransformerEncoderBlock.forward (:130-161).  The first layer_norm sees a rank-2 tensor (per-row
    moments), the second a rank-3 one (moments over frames AND units jointly); there is no second residual (:156)."""
    pre = scope + "/" if scope else ""
    att = attention_modules_mha(x2d, params, scope, num_heads, num_units, max_frames, block_id)       # :133-135
    att = att.reshape(-1, num_units * num_heads)                                                       # :138
    att = torch.relu(att @ params[pre + "dense/kernel"] + params[pre + "dense/bias"])                  # :141
    att = layer_norm(att + x2d, params, pre + "LayerNorm")                                             # :145-146
    out = att.reshape(-1, max_frames, feature_size)                                                    # :149
    out = torch.relu(out @ params[pre + "conv1d/kernel"][0] + params[pre + "conv1d/bias"])             # :150-151
    out = out @ params[pre + "conv1d_1/kernel"][0] + params[pre + "conv1d_1/bias"]                     # :152
    out = layer_norm(out, params, pre + "LayerNorm_1")                                                 # :155
    return out.reshape(-1, feature_size)                                                               # :156


# --------------------------------------------------------------------------------------
# a11-a13: MoeModel, CrossEntropyLoss, regulariser
# --------------------------------------------------------------------------------------
def moe_forward(act, params, vocab_size, num_mixtures, cfg: Optional["OracleConfig"] = None, is_training=True, updates=None):
    """MoeModel.create_model (video_level_models.py:85-158).  Default branch: one bias-free gates layer.  With cfg:
    moe_low_rank_gating > 0 -> gates = (act gates1) gates2, both bias-free (:94-108); moe_prob_gating -> the class probabilities are
    gated by sigmoid(BN(p W)) ('prob' input, W [V, V]) or sigmoid(BN(act W)) (W [H, V]), optionally without W's diagonal (:128-156)."""
    if "gates1/weights" in params:
        gate = (act @ params["gates1/weights"]) @ params["gates2/weights"]   # :94-108
    else:
        gate = act @ params["gates/weights"]                                # no bias :86-93
    expert = act @ params["experts/weights"] + params["experts/biases"]      # :109-114
    gating = torch.softmax(gate.reshape(-1, num_mixtures + 1), dim=-1)       # :116-118
    experts = torch.sigmoid(expert.reshape(-1, num_mixtures))                # :119-121
    probs = (gating[:, :num_mixtures] * experts).sum(dim=1)                  # :123-124
    probs = probs.reshape(-1, vocab_size)                                    # :125-126
    if cfg is not None and cfg.moe_prob_gating:
        W = params["gating_prob_weights"]
        gates = (probs if cfg.moe_prob_gating_input == "prob" else act) @ W  # :129-142
        if cfg.remove_diag:
            gates = gates - torch.diagonal(W) * probs                       # :144-147
        gates = batch_norm(gates, params, "gating_prob_bn", is_training, updates)   # :149-154
        probs = probs * torch.sigmoid(gates)                                 # :156-158
    return probs


def cross_entropy_loss(predictions, labels):
    """CrossEntropyLoss.calculate_loss, epsilon = 10e-6 (losses.py:44-51)."""
    eps = 10e-6
    y = labels.to(predictions.dtype)
    ce = y * torch.log(predictions + eps) + (1 - y) * torch.log(1 - predictions + eps)
    return (-ce).sum(dim=1).mean()


def regularization_loss(params, cfg: OracleConfig):
    """slim.l2_regularizer(moe_l2) on the two MoE FC weights: s * sum(w^2)/2
    (video_level_models.py:91,113; collected at train.py:301-303)."""
    gate_ws = [n for n in ("gates/weights", "gates1/weights", "gates2/weights") if n in params]          # :91,99,106
    reg = cfg.moe_l2 * 0.5 * (sum((params[n] ** 2).sum() for n in gate_ws) + (params["experts/weights"] ** 2).sum())
    if cfg.model == "WillowModelReg":       # orthogonal_regularizer on both cluster_weights2 (video_pooling_modules.py:1561-1568)
        reg = reg + orthogonal_regularizer(params["video_VLAD/cluster_weights2"], cfg.rgb_det_reg)
        if "audio_VLAD/cluster_weights2" in params:
            reg = reg + orthogonal_regularizer(params["audio_VLAD/cluster_weights2"], cfg.audio_det_reg)
    return reg


# --------------------------------------------------------------------------------------
# a2-a11 assembled: NetVladV1 / NetVladV2 .create_model
# --------------------------------------------------------------------------------------
def model_forward(params, model_input, num_frames, cfg: OracleConfig, is_training=True,
                  updates=None, dropout_masks=None, return_intermediates=False):
    """NetVladV1.create_model (frame_level_models.py:2224-2377) / NetVladV2 (:2385-2513).
    model_input [B, max_frames, 1152|1024] already L2-normalised per frame by the caller
    (train.py:262-264).  Returns predictions [B, vocab]."""
    inter = {}
    S, K = cfg.iterations, cfg.cluster_size
    if cfg.model == "WillowModelReg":       # :2539-2544; the uniform draws come in through dropout_masks["frame_uniform"]
        u = (dropout_masks or {})["frame_uniform"]
        x = (sample_random_frames if cfg.sample_random_frames else sample_random_sequence)(model_input, num_frames, S, u)
    else:
        x = sample_uniform_frames(model_input, num_frames, S)               # :2255
    feat = x.shape[2]
    x2d = x.reshape(-1, feat)                                               # :2259
    if cfg.add_batch_norm:
        x2d = batch_norm(x2d, params, "input_bn", is_training, updates)     # :2265-2271
    inter["input_bn"] = x2d
    has_audio = feat > cfg.video_dim                                        # App. C9
    xv = x2d[:, :cfg.video_dim]
    xa = x2d[:, cfg.video_dim:]
    Ka = K // 4                                                             # App. C8
    dm = dropout_masks or {}
    if cfg.model == "NetVladV1":
        vv = netvlad_forward(xv, params, "video_VLAD", S, cfg.add_batch_norm, is_training, updates)   # :2273-2274
        inter["vlad_video"] = vv
        if has_audio:
            va = netvlad_forward(xa, params, "audio_VLAD", S, cfg.add_batch_norm, is_training, updates)  # :2276-2277
            inter["vlad_audio"] = va
        if cfg.encoder:
            B = vv.shape[0]
            # App. C5: tokens = clusters.  [B, D*K] d-major -> [B,K,D]
            tv = vv.reshape(B, cfg.video_dim, K).transpose(1, 2)
            vv = transformer_encoder(tv, params, "video_attention", 64, "encode1").reshape(B, -1)     # :2282-2292
            if has_audio:
                ta = va.reshape(B, cfg.audio_dim, Ka).transpose(1, 2)
                va = transformer_encoder(ta, params, "audio_attention", 16, "encode2").reshape(B, -1)  # :2294-2304
    elif cfg.model == "WillowModelReg":
        vv = netvlad_orthoreg_forward(xv, params, "video_VLAD", "netvlad_rgb_scope", S, cfg.add_batch_norm, is_training, updates)
        inter["vlad_video"] = vv
        if has_audio:
            va = netvlad_orthoreg_forward(xa, params, "audio_VLAD", "netvlad_audio_scope", S, cfg.add_batch_norm, is_training,
                                          updates)
            inter["vlad_audio"] = va
    elif cfg.model == "NetVladV2":
        vv = netvlad_atten_cluster_forward(xv, params, "video_VLAD", S, is_training, cfg.v2_dropout_rate,
                                           dm.get("video"), updates)       # :2437-2438
        inter["vlad_video"] = vv
        if has_audio:
            va = netvlad_atten_cluster_forward(xa, params, "audio_VLAD", S, is_training, cfg.v2_dropout_rate,
                                               dm.get("audio"), updates)   # :2440-2441
            inter["vlad_audio"] = va
    else:
        raise ValueError(cfg.model)
    vlad = torch.cat([vv, va], dim=1) if has_audio else vv                  # :2309 / :2445
    inter["vlad"] = vlad
    act = vlad @ params["hidden1_weights"]                                  # :2319
    if cfg.add_batch_norm and cfg.relu:
        act = batch_norm(act, params, "hidden1_bn", is_training, updates)   # :2321-2327
    else:
        act = act + params["hidden1_biases"]                                # :2329-2334
    if cfg.relu:
        act = torch.clamp(act, 0.0, 6.0)                                    # relu6 :2336-2337
    if cfg.gating:
        G = params["gating_weights_2"]
        gates = act @ G                                                     # :2347
        if cfg.remove_diag:
            gates = gates - torch.diagonal(G) * act                        # :2349-2352
        gates = batch_norm(gates, params, "gating_bn", is_training, updates)  # :2354-2360 (App. C12)
        act = act * torch.sigmoid(gates)                                    # :2367-2368
    inter["activation"] = act
    pred = moe_forward(act, params, cfg.vocab_size, cfg.moe_num_mixtures, cfg, is_training, updates)   # :2370-2377
    if return_intermediates:
        return pred, inter
    return pred


# --------------------------------------------------------------------------------------
# Parameter construction
# --------------------------------------------------------------------------------------
def _bn(p, scope, n, dtype):
    p[scope + "/beta"] = torch.zeros(n, dtype=dtype)
    p[scope + "/gamma"] = torch.ones(n, dtype=dtype)
    p[scope + "/moving_mean"] = torch.zeros(n, dtype=dtype)
    p[scope + "/moving_variance"] = torch.ones(n, dtype=dtype)


def _glorot(gen, fan_in, fan_out, dtype):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(fan_in, fan_out, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


def _normal(gen, shape, std, dtype):
    return (torch.randn(*shape, generator=gen, dtype=torch.float64) * std).to(dtype)


def init_params(cfg: OracleConfig, feature_size: int = 1152, seed: int = 1000,
                dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Random-init weights with the reference's shapes, names and initialiser families."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}
    K, S, H, V, m = cfg.cluster_size, cfg.iterations, cfg.hidden_size, cfg.vocab_size, cfg.moe_num_mixtures
    streams = [("video", cfg.video_dim, K, 64, "encode1")]
    if feature_size > cfg.video_dim:
        streams.append(("audio", cfg.audio_dim, K // 4, 16, "encode2"))
    _bn(p, "input_bn", feature_size, dtype)
    vlad_dim = 0
    for name, D, Ks, heads, sid in streams:
        sc = f"{name}_VLAD"
        vlad_dim += D * Ks
        if cfg.model == "WillowModelReg":
            sid2 = "netvlad_rgb_scope" if name == "video" else "netvlad_audio_scope"
            p[f"{sc}/cluster_weights{sid2}"] = _normal(g, (D, Ks), 1 / math.sqrt(D), dtype)
            _bn(p, sc + "/cluster_bn", Ks, dtype)
            p[f"{sc}/cluster_biases{sid2}"] = _normal(g, (Ks,), 1 / math.sqrt(D), dtype)
            p[sc + "/cluster_weights2"] = _normal(g, (D, Ks), 1 / math.sqrt(D), dtype)
        elif cfg.model == "NetVladV1":
            p[sc + "/cluster_weights"] = _normal(g, (D, Ks), 1 / math.sqrt(D), dtype)
            _bn(p, sc + "/cluster_bn", Ks, dtype)
            p[sc + "/cluster_biases"] = _normal(g, (Ks,), 1 / math.sqrt(D), dtype)
            p[sc + "/cluster_weights2"] = _normal(g, (1, D, Ks), 1 / math.sqrt(D), dtype)
            if cfg.encoder:
                a = f"{name}_attention"
                for w in ("q", "k", "v"):
                    p[f"{a}/{w}/kernel"] = _glorot(g, D, D, dtype)
                p[a + "/output_transform/kernel"] = _glorot(g, D, D, dtype)
                p[a + "/output_transform/bias"] = torch.zeros(D, dtype=dtype)
                p[a + f"/filter_output{sid}/kernel"] = _glorot(g, D, 4 * D, dtype)
                p[a + f"/filter_output{sid}/bias"] = torch.zeros(4 * D, dtype=dtype)
                p[a + f"/ff_output{sid}/kernel"] = _glorot(g, 4 * D, D, dtype)
                p[a + f"/ff_output{sid}/bias"] = torch.zeros(D, dtype=dtype)
                for ln in ("LayerNorm", "LayerNorm_1", "LayerNorm_2"):
                    p[f"{a}/{ln}/beta"] = torch.zeros(D, dtype=dtype)
                    p[f"{a}/{ln}/gamma"] = torch.ones(D, dtype=dtype)
        else:
            a = sc + "/cluster_attention"
            for w in ("q", "k", "v"):
                p[f"{a}/{w}/kernel"] = _glorot(g, D, D, dtype)
            _bn(p, a + "/logits_bn", S, dtype)
            _bn(p, a + "/attention_bn", D, dtype)
            p[a + "/output_transform/kernel"] = _glorot(g, D, D, dtype)
            p[a + "/output_transform/bias"] = torch.zeros(D, dtype=dtype)
            p[a + "/LayerNorm/beta"] = torch.zeros(D, dtype=dtype)
            p[a + "/LayerNorm/gamma"] = torch.ones(D, dtype=dtype)
            p[a + "/filter_outputencode/kernel"] = _glorot(g, D, 4 * D, dtype)
            p[a + "/filter_outputencode/bias"] = torch.zeros(4 * D, dtype=dtype)
            _bn(p, a + "/filter_bn", 4 * D, dtype)
            p[a + "/ff_outputencode/kernel"] = _glorot(g, 4 * D, Ks, dtype)
            p[a + "/ff_outputencode/bias"] = torch.zeros(Ks, dtype=dtype)
            _bn(p, a + "/feed_output_bn", Ks, dtype)
            p[sc + "/cluster_centers"] = _normal(g, (D, Ks), 1 / math.sqrt(D), dtype)
    p["hidden1_weights"] = _normal(g, (vlad_dim, H), 1 / math.sqrt(K), dtype)
    if cfg.add_batch_norm and cfg.relu:
        _bn(p, "hidden1_bn", H, dtype)
    else:
        p["hidden1_biases"] = _normal(g, (H,), 0.01, dtype)
    if cfg.gating:
        p["gating_weights_2"] = _normal(g, (H, H), 1 / math.sqrt(H), dtype)
        _bn(p, "gating_bn", H, dtype)
    if cfg.moe_low_rank_gating != -1:
        p["gates1/weights"] = _glorot(g, H, cfg.moe_low_rank_gating, dtype)
        p["gates2/weights"] = _glorot(g, cfg.moe_low_rank_gating, V * (m + 1), dtype)
    else:
        p["gates/weights"] = _glorot(g, H, V * (m + 1), dtype)
    p["experts/weights"] = _glorot(g, H, V * m, dtype)
    p["experts/biases"] = torch.zeros(V * m, dtype=dtype)
    if cfg.moe_prob_gating:
        rows = V if cfg.moe_prob_gating_input == "prob" else H
        p["gating_prob_weights"] = _normal(g, (rows, V), 1 / math.sqrt(V), dtype)      # :131-140
        _bn(p, "gating_prob_bn", V, dtype)
    return p


def trainable_names(params, cfg: Optional[OracleConfig] = None) -> List[str]:
    """Everything but BN moving statistics; cluster_biases only when BN is off."""
    out = []
    for n in params:
        if n.endswith("/moving_mean") or n.endswith("/moving_variance"):
            continue
        if "/cluster_biases" in n and (cfg is None or cfg.add_batch_norm):
            continue
        out.append(n)
    return out


# --------------------------------------------------------------------------------------
# a13-a15: loss assembly, tower combine, clip, LR, Adam (train.py:244-336; utils.py:170-213)
# --------------------------------------------------------------------------------------
def loss_and_grads(params, model_input, num_frames, labels, cfg, dropout_masks=None):
    """One tower: final_loss = regularization_penalty * reg + label_loss (train.py:294-323)."""
    names = trainable_names(params, cfg)
    leaf = dict(params)
    for n in names:
        leaf[n] = params[n].detach().clone().requires_grad_(True)
    updates: Dict[str, torch.Tensor] = {}
    pred = model_forward(leaf, model_input, num_frames, cfg, True, updates, dropout_masks)
    label_loss = cross_entropy_loss(pred, labels)
    final = cfg.regularization_penalty * regularization_loss(leaf, cfg) + label_loss
    grads = torch.autograd.grad(final, [leaf[n] for n in names], allow_unused=True)
    gd = {n: (g if g is not None else torch.zeros_like(leaf[n])) for n, g in zip(names, grads)}
    return pred.detach(), label_loss.detach(), gd, updates


def combine_gradients(tower_grads: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """SUM (not mean) over towers (utils.py:192-213)."""
    return {n: torch.stack([tg[n] for tg in tower_grads], 0).sum(0) for n in tower_grads[0]}


def clip_gradient_norms(grads: Dict[str, torch.Tensor], max_norm: float) -> Dict[str, torch.Tensor]:
    """Per-variable tf.clip_by_norm: t * c / max(||t||, c) (utils.py:170-189)."""
    out = {}
    for n, g in grads.items():
        nrm = torch.sqrt((g * g).sum())
        out[n] = g * (max_norm / torch.clamp(nrm, min=max_norm))
    return out


def learning_rate(cfg: OracleConfig, global_step: int, batch_size: int, num_towers: int) -> float:
    """tf.train.exponential_decay(..., staircase=True) on step*batch*towers examples (train.py:244-249)."""
    p = math.floor(global_step * batch_size * num_towers / cfg.learning_rate_decay_examples)
    return cfg.base_learning_rate * cfg.learning_rate_decay ** p


def adam_tf_update(param, grad, m, v, lr: float, t: int, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t * m / (sqrt(v) + eps)
    (epsilon on the un-bias-corrected sqrt(v); train.py:252,336)."""
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    return param - lr_t * m / (torch.sqrt(v) + eps), m, v


def train_step(params, opt_state, model_input, num_frames, labels, cfg: OracleConfig,
               num_towers: int = 1, dropout_masks=None):
    """One optimiser step as train.build_graph wires it (train.py:266-336): split the global
    batch over towers, per-tower grads, SUM, per-variable clip, Adam; BN moving averages
    are updated from every tower in tower order.  ``opt_state`` = {"step": int, "m": {}, "v": {}}.
    Returns (new_params, new_opt_state, info)."""
    B = model_input.shape[0]
    assert B % num_towers == 0
    per = B // num_towers
    tower_grads, losses, preds = [], [], []
    new_params = dict(params)
    for i in range(num_towers):
        sl = slice(i * per, (i + 1) * per)
        dmi = None if dropout_masks is None else {k: v[sl] for k, v in dropout_masks.items()}
        pred, loss, gd, upd = loss_and_grads(params, model_input[sl], num_frames[sl], labels[sl], cfg, dmi)
        tower_grads.append(gd); losses.append(loss); preds.append(pred)
        for n, val in upd.items():
            new_params[n] = new_params[n] * BN_DECAY + val.to(new_params[n].dtype) * (1 - BN_DECAY)
    merged = combine_gradients(tower_grads)                                 # train.py:330
    if cfg.clip_gradient_norm > 0:
        merged = clip_gradient_norms(merged, cfg.clip_gradient_norm)        # :332-334
    step = opt_state["step"]
    lr = learning_rate(cfg, step, per, num_towers)                          # :244-249 (batch_size is per tower)
    t = step + 1
    new_m, new_v = {}, {}
    for n, g in merged.items():
        m0 = opt_state["m"].get(n, torch.zeros_like(params[n]))
        v0 = opt_state["v"].get(n, torch.zeros_like(params[n]))
        new_params[n], new_m[n], new_v[n] = adam_tf_update(params[n], g, m0, v0, lr, t)
    info = {"loss": torch.stack(losses).mean(), "predictions": torch.cat(preds, 0), "clipped_grads": merged, "lr": lr,
            "adam_m": new_m, "adam_v": new_v}
    return new_params, {"step": t, "m": new_m, "v": new_v}, info


# --------------------------------------------------------------------------------------
# Synthetic, reader-faithful input
# --------------------------------------------------------------------------------------
def make_synthetic_batch(batch: int, max_frames: int = 300, feature_size: int = 1152, vocab_size: int = 3862,
                         seed: int = 0, min_frames: Optional[int] = None, dtype=torch.float32):
    """uint8 ~ U{0..255} -> Dequantize q*4/255 + 4/512 - 2 (utils.py:28-43) -> zero frames >= num_frames
    (readers.py:189-193) -> per-frame L2 normalise (train.py:262-264); 1-5 positive labels per clip."""
    rng = np.random.Generator(np.random.PCG64(seed))
    q = rng.integers(0, 256, size=(batch, max_frames, feature_size), dtype=np.uint8)
    lo = min_frames if min_frames is not None else max(1, (max_frames * 2) // 5)
    nf = rng.integers(lo, max_frames + 1, size=(batch,), dtype=np.int64)
    raw = q.astype(np.float32) * np.float32(4.0 / 255.0) + np.float32(4.0 / 512.0 - 2.0)
    mask = (np.arange(max_frames)[None, :] < nf[:, None])[..., None]
    raw = raw * mask
    labels = np.zeros((batch, vocab_size), dtype=bool)
    for b in range(batch):
        npos = int(rng.integers(1, 6))
        labels[b, rng.choice(vocab_size, size=npos, replace=False)] = True
    x = torch.from_numpy(raw).to(dtype)
    x = l2_normalize(x, 2)
    return x, torch.from_numpy(nf.astype(np.int32)), torch.from_numpy(labels)
