"""NetVladOrthoReg and NetVladAttenCluster (reference: video_pooling_modules.py:1499-1586, 1589-1663)."""
from __future__ import annotations

import math

from . import layers, module_utils, modules, ops, transformer_utils
from . import variables as vs


class NetVladOrthoReg(modules.BaseModule):
    """NetVLAD from WILLOW's model with orthogonal regularisation (video_pooling_modules.py:1499-1586): the pooling is
    NetVLAD's (same K1/K2/K3 kernels) with a 2-D ``cluster_weights2`` [D, K] and scope-id-suffixed variable names; the
    penalty ``det_reg * sum |W2n^T W2n - I|`` is collected as a regularisation loss."""

    def __init__(self, feature_size, max_frames, cluster_size, batch_norm, is_training, det_reg=None, scope_id=None):
        self.feature_size = feature_size
        self.max_frames = max_frames
        self.is_training = is_training
        self.batch_norm = batch_norm
        self.cluster_size = int(cluster_size)
        self.det_reg = det_reg
        self.scope_id = scope_id

    def forward(self, inputs, **unused_params):
        D, K, dev = self.feature_size, self.cluster_size, inputs.device
        sid = "" if self.scope_id is None else str(self.scope_id)
        std = 1 / math.sqrt(D)
        cluster_weights = vs.get_variable("cluster_weights" + sid, [D, K], vs.random_normal_initializer(std), device=dev)
        bn = bias = None
        if self.batch_norm:
            bn = layers.bn_variables("cluster_bn", K, dev)
        else:
            bias = vs.get_variable("cluster_biases" + sid, [K], vs.random_normal_initializer(std), device=dev)
        cluster_weights2 = vs.get_variable("cluster_weights2", [D, K], vs.random_normal_initializer(std), device=dev)
        if self.det_reg is not None:
            reg = module_utils.orthogonal_regularizer(self.det_reg, self.scope_id)(cluster_weights2)
            if reg is not None:
                vs.default_store().add_regularization_loss(reg)
        return ops.netvlad(inputs, cluster_weights, cluster_weights2.reshape(1, D, K), self.max_frames, bn=bn, bias=bias,
                           is_training=self.is_training)


class NetVladAttenCluster(modules.BaseModule):
    """NetVLAD whose cluster similarities come from a frame-level transformer encoder."""

    def __init__(self, feature_size, max_frames, cluster_size, batch_norm, is_training, scope_id=None):
        self.feature_size = feature_size
        self.max_frames = max_frames
        self.is_training = is_training
        self.batch_norm = batch_norm
        self.cluster_size = int(cluster_size)
        self.scope_id = scope_id
        self.encoder_hidden_size = feature_size
        self.num_heads = feature_size // 16               # :1613
        self.dropout_ratio = 0.1
        self.filter_size = 4 * self.encoder_hidden_size   # :1615

    def forward(self, inputs, dropout_mask=None, dropout_rate=None, lazy=False, **unused_params):
        """inputs [(B*max_frames), F] -> [B, F*K] (f-major), L2-normalised (lazy: the lazily normalised form, ops.vlad_aggregate)."""
        reshaped_input = inputs.reshape(-1, self.max_frames, self.feature_size)          # :1623
        with vs.variable_scope("cluster_attention"):
            encoder_block = transformer_utils.TransformerEncoderMod(
                feature_size=self.feature_size, hidden_size=self.encoder_hidden_size, num_heads=self.num_heads,
                attention_dropout=self.dropout_ratio, ff_filter_size=self.filter_size, ff_relu_dropout=0.1,
                is_train=self.is_training, scope_id="encode", final_size=self.cluster_size)
            # (the frames are read twice, by the encoder and by the aggregation below: their two gradients meet inside the encoder's
            # backward instead of in an add pass of autograd's -- ops.GradJoin)
            join = ops.GradJoin() if (self.is_training and inputs.is_cuda) else None
            cluster_similarities = encoder_block.forward(reshaped_input, dropout_mask=dropout_mask,
                                                         dropout_rate=dropout_rate, grad_join=join)          # [B,S,K] :1638
        cluster_centres = vs.get_variable("cluster_centers", [self.feature_size, self.cluster_size],
                                          vs.random_normal_initializer(1 / math.sqrt(self.feature_size)),
                                          device=inputs.device)                                # :1641-1643
        # sum_n sims * (x - c), intra-L2, flatten, L2 (:1646-1658; App. C6/C7) -- HIP kernel K2
        return ops.vlad_aggregate(cluster_similarities, inputs, cluster_centres, self.max_frames, lazy=lazy, grad_join=join)
