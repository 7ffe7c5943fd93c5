"""Attention-cluster pooling and the per-head transformer block of the reference's attention_modules.py:22-161
(SURVEY.md 8(f) rank 4), on the hot path's kernels:

* ``OneFcAttention``: the weighted frame sums are K2's aggregation (similarities given, no softmax over clusters) and their
  backward is K3.  Because the attention is a softmax over the FRAMES of a clip, every cluster's weights sum to one, so the
  shift ``alpha * act + beta`` is exactly K2's residual term with the constant centre ``-beta / alpha``
  (``sum_t a (x - c) = act - c``) up to the sign of alpha, and ``l2_normalize(., 1) / sqrt(K)`` is K2's intra-normalisation
  followed by its global L2 (K unit-norm clusters have global norm sqrt(K)).  The k-major descriptor layout is the
  reference's ``[B, num_cluster * num_feature]``.
* ``MultiHeadAttention``: every head owns its relu(dense) q / k / v; the per-head cores run as ONE K4 launch over the
  concatenated heads when the head width is one K4 covers (8 or 16), as library batched GEMMs otherwise.
* ``TransformerEncoderBlock``: dense / conv1d(kernel_size=1) layers through ``layers.dense`` (split-bf16 library GEMMs).

Variable names are the ones TF1 assigns by default (``dense``, ``dense_1``, ``conv1d``, ``LayerNorm_1`` ...) for the first
block built under a variable scope; build further blocks under their own ``variable_scope``."""
from __future__ import annotations

import math

import torch

from . import layers, modules, ops
from . import variables as vs


def _constant_initializer(value: float):
    return lambda shape, dev, gen: torch.full(tuple(shape), float(value), device=dev)


class OneFcAttention(modules.BaseModule):
    """attention_modules.py:22-64."""

    def __init__(self, num_features, num_frames, num_cluster, do_shift=True):
        self.num_feature = int(num_features)
        self.num_frames = int(num_frames)
        self.num_cluster = int(num_cluster)
        self.do_shift = do_shift

    def forward(self, inputs, **unused_params):
        """inputs [(B * num_frames), F] -> [B, num_cluster * F] (cluster-major)."""
        F_, T, K = self.num_feature, self.num_frames, self.num_cluster
        attention_weights = vs.get_variable("one_fc_attention_weight", [F_, K], vs.glorot_uniform_initializer(),
                                            device=inputs.device)                                   # :30-33
        rows = inputs.shape[0]
        if layers.use_split_gemm(inputs, rows, K):
            attention = ops.dense_x3(inputs, attention_weights)                                     # :34
        else:
            attention = inputs.matmul(attention_weights)
        attention = torch.softmax(attention.reshape(-1, T, K) * (1.0 / math.sqrt(F_)), dim=1)     # :35-37 (over frames)
        if not self.do_shift:
            activation = torch.bmm(attention.transpose(1, 2), inputs.reshape(-1, T, F_))           # :39-41
            return activation.reshape(-1, K * F_)
        alpha = vs.get_variable("alpha", [1], _constant_initializer(1.0), device=inputs.device)    # :47-50
        beta = vs.get_variable("beta", [1], _constant_initializer(0.01), device=inputs.device)     # :51-54
        # :56-59 as K2's residual + intra-L2 + global L2 (module docstring); alpha == 0 has no such form and gives inf / nan
        centres = (-(beta / alpha)).expand(F_, K).contiguous()
        pooled = ops.vlad_aggregate(attention, inputs, centres, T, kmajor=True)                     # [B, K, F]
        return (pooled * torch.sign(alpha)).reshape(-1, K * F_)                                     # :61


class MultiHeadAttention(modules.BaseModule):
    """attention_modules.py:67-112."""

    def __init__(self, num_heads, num_units, max_frames, block_id):
        self.num_heads = int(num_heads)
        self.num_units = int(num_units)
        self.max_frames = int(max_frames)
        self.block_id = block_id

    def _projections(self, inputs, scope_id):
        with vs.variable_scope("Block{}Layer{}".format(self.block_id, scope_id)):                   # :79
            return [layers.dense(inputs, self.num_units, use_bias=True, name=name, activation=torch.relu)
                    .reshape(-1, self.max_frames, self.num_units) for name in ("dense", "dense_1", "dense_2")]   # :81-89

    def forward(self, inputs, **unused_params):
        """inputs [(B * max_frames), F] -> [B, max_frames, num_units * num_heads]."""
        heads = [self._projections(inputs, i) for i in range(self.num_heads)]
        q, k, v = (torch.cat([h[j] for h in heads], dim=2) for j in range(3))
        scale = 1.0 / float(self.num_units)                      # logits / sqrt(u) / sqrt(u)  (:95-96)
        if q.is_cuda and self.num_units in (8, 16) and self.max_frames <= 512:
            return ops.mha_core(q, k, v, self.num_heads, scale)                                      # :92-98, all heads at once
        B, L, h, u = q.shape[0], self.max_frames, self.num_heads, self.num_units
        qh, kh, vh = (t.reshape(B, L, h, u).permute(0, 2, 1, 3) for t in (q, k, v))
        att = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)) * scale, dim=-1)
        return torch.matmul(att, vh).permute(0, 2, 1, 3).reshape(B, L, h * u)                        # :104-110


def _conv1d_k1(x, filters, name, activation=None):
    """tf.layers.conv1d(kernel_size=1, use_bias=True): a dense layer whose kernel variable is [1, in, out]."""
    with vs.variable_scope(name):
        g = vs.glorot_uniform_initializer()
        kernel = vs.get_variable("kernel", [1, x.shape[-1], filters],
                                 lambda shape, dev, gen: g(shape[1:], dev, gen).reshape(shape), device=x.device)
        bias = vs.get_variable("bias", [filters], vs.zeros_initializer(), device=x.device)
    rows = x.numel() // x.shape[-1]
    if layers.use_split_gemm(x, rows, filters):
        y = ops.dense_x3(x.reshape(rows, x.shape[-1]), kernel[0]).reshape(*x.shape[:-1], filters)
    else:
        y = x.matmul(kernel[0])
    y = y + bias
    return activation(y) if activation is not None else y


class TransformerEncoderBlock(modules.BaseModule):
    """attention_modules.py:115-161."""

    def __init__(self, is_training, num_units, max_frames, feature_size, num_heads, block_id):
        self.is_training = is_training
        self.num_units = int(num_units)
        self.max_frames = int(max_frames)
        self.feature_size = int(feature_size)
        self.num_heads = int(num_heads)
        self.block_id = block_id

    def forward(self, inputs, **unused_params):
        """inputs [(B * max_frames), feature_size] -> same shape (num_units must equal feature_size, :156)."""
        multi_head_layer = MultiHeadAttention(self.num_heads, self.num_units, self.max_frames, self.block_id)
        attention_output = multi_head_layer.forward(inputs)                                          # :133-135
        attention_output = attention_output.reshape(-1, self.num_units * self.num_heads)             # :138
        attention_output = layers.dense(attention_output, self.feature_size, use_bias=True, name="dense",
                                        activation=torch.relu)                                       # :141
        attention_output = layers.layer_norm(attention_output + inputs, "LayerNorm")                 # :145-146 (rank 2: per row)
        output = attention_output.reshape(-1, self.max_frames, self.feature_size)                    # :149
        output = _conv1d_k1(output, 4 * self.num_units, "conv1d", activation=torch.relu)             # :150-151
        output = _conv1d_k1(output, self.num_units, "conv1d_1")                                      # :152
        output = layers.layer_norm(output, "LayerNorm_1")                                            # :155 (rank 3: frames x units)
        return output.reshape(-1, self.feature_size)                                                 # :156
