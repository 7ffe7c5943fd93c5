"""Transformer blocks used by NetVladV1 / NetVladV2 (reference: transformer_utils.py:374-457, 507-766).
The attention cores (QK^T -> softmax -> .V) run in the HIP kernel K4 (csrc/mha.hip); the dense
projections are library GEMMs."""
from __future__ import annotations

import torch

from . import layers, modules, ops
from . import variables as vs


class MultiHeadAttention(modules.BaseModule):
    """transformer_utils.py:507-586."""

    def __init__(self, feature_size, hidden_size, num_heads, attention_dropout, is_train):
        self.feature_size = feature_size
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.attention_dropout = attention_dropout
        self.is_train = is_train

    def forward(self, queries, keys, defer_bias=False):
        """defer_bias: return (output without the output_transform bias, bias) -- the caller's fused residual layer_norm adds it."""
        q, k, v = layers.qkv_projections(queries, keys, self.hidden_size)          # :559-561
        depth = self.hidden_size // self.num_heads
        # split_heads, q *= depth**-0.5, softmax(q k^T) v, combine_heads (:564-581): one kernel
        attention_output = ops.mha_core(q, k, v, self.num_heads, depth ** -0.5)
        return layers.dense(attention_output, self.feature_size, use_bias=True, name="output_transform",
                            defer_bias=defer_bias)                                 # :583


class MultiHeadAttentionBN(modules.BaseModule):
    """transformer_utils.py:589-677: no q scaling, batch_norm on the logits and on the combined heads."""

    def __init__(self, feature_size, hidden_size, num_heads, attention_dropout, is_train):
        self.feature_size = feature_size
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.attention_dropout = attention_dropout
        self.is_train = is_train

    def forward(self, queries, keys, defer_bias=False):
        """defer_bias: return (raw output of output_transform, its bias) where the fused node allows it -- the caller's layer norm adds
        the bias (and applies the dropout between the two) in its own passes -- else (output, None)."""
        L = keys.shape[1]
        rows = queries.numel() // queries.shape[-1]
        if (queries is keys and self.is_train and layers.use_split_gemm(queries, rows, self.hidden_size)
                and ops.qkv_attention_bn_ok(queries, self.hidden_size, self.num_heads)):
            # q/k/v projections + logits_bn attention as ONE node, so that its backward hands [dq | dk | dv] to the projections' GEMMs as
            # their operand image (same variables, same order: q, k, v kernels, then logits_bn's)
            wq, _ = layers.dense_variables("q", queries.shape[-1], self.hidden_size, False, queries.device)
            wk, _ = layers.dense_variables("k", queries.shape[-1], self.hidden_size, False, queries.device)
            wv, _ = layers.dense_variables("v", queries.shape[-1], self.hidden_size, False, queries.device)
            gamma, beta, mm, mv = layers.bn_variables("logits_bn", L, queries.device)    # channel = key position :652-658
            attention_output = ops.qkv_attention_bn_x3(queries, wq, wk, wv, gamma, beta, mm, mv, self.num_heads)
        else:
            q, k, v = layers.qkv_projections(queries, keys, self.hidden_size)
            gamma, beta, mm, mv = layers.bn_variables("logits_bn", L, q.device)        # channel = key position :652-658
            attention_output = ops.mha_core_bn(q, k, v, self.num_heads, gamma, beta, mm, mv, self.is_train)
        rows = attention_output.numel() // attention_output.shape[-1]
        if (self.is_train and layers.use_split_gemm(attention_output, rows, self.feature_size)
                and ops.bn_dense_x3_ok(attention_output, self.feature_size)):
            # attention_bn -> output_transform as ONE node: the normalised tensor leaves the batch norm only as the GEMM's operand image
            # (variables in the unfused order: attention_bn's, then the dense layer's)
            g2, b2, mm2, mv2 = layers.bn_variables("attention_bn", attention_output.shape[-1], attention_output.device)
            W, bias = layers.dense_variables("output_transform", attention_output.shape[-1], self.feature_size, True, attention_output.device)
            if defer_bias:
                return ops.bn_dense_x3(attention_output, g2, b2, mm2, mv2, W), bias
            out = ops.bn_dense_x3(attention_output, g2, b2, mm2, mv2, W, bias=bias)
            return out
        attention_output = layers.batch_norm(attention_output, self.is_train, "attention_bn")   # :666-671
        out = layers.dense(attention_output, self.feature_size, use_bias=True, name="output_transform")
        return (out, None) if defer_bias else out


class FeedForwardNetwork(modules.BaseModule):
    """transformer_utils.py:679-715 (relu on both dense layers, residual + layer_norm inside)."""

    def __init__(self, feature_size, filter_size, relu_dropout, is_train, scope_id):
        self.feature_size = feature_size
        self.filter_size = filter_size
        self.relu_dropout = relu_dropout
        self.is_train = is_train
        self.scope_id = scope_id

    def forward(self, inputs, **unused_params):
        n1, n2 = "filter_output{}".format(self.scope_id), "ff_output{}".format(self.scope_id)
        rows = inputs.numel() // inputs.shape[-1]
        if layers.use_split_gemm(inputs, rows, self.filter_size) and self.feature_size % 8 == 0:
            # both dense layers as split-bf16 library GEMMs with the inner bias + ReLU fused into the operand split
            w1, b1 = layers.dense_variables(n1, inputs.shape[-1], self.filter_size, True, inputs.device)
            w2, b2 = layers.dense_variables(n2, self.filter_size, self.feature_size, True, inputs.device)
            output = ops.ffn_x3(inputs.reshape(rows, inputs.shape[-1]), w1, b1, w2).reshape(*inputs.shape[:-1], self.feature_size)
            # relu(output + b2) + inputs, then layer_norm (:708-713): one fused kernel pair
            return layers.layer_norm(output, "LayerNorm_1", residual=inputs, bias=b2, relu=True)
        else:
            filter_output = layers.dense(inputs, self.filter_size, True, n1, torch.relu)                       # :701-704
            output = layers.dense(filter_output, self.feature_size, True, n2, torch.relu)                      # :708-711
        return layers.layer_norm(output, "LayerNorm_1", residual=inputs)      # output + inputs, then layer_norm :712-713


class FeedForwardNetworkMod(modules.BaseModule):
    """transformer_utils.py:718-766."""

    def __init__(self, feature_size, filter_size, relu_dropout, is_train, scope_id, final_size):
        self.feature_size = feature_size
        self.filter_size = filter_size
        self.relu_dropout = relu_dropout
        self.is_train = is_train
        self.scope_id = scope_id
        self.final_size = final_size

    def first_kernel(self):
        """The first dense layer's kernel if it exists already (never created here: variable creation order is the reference's) -- the
        layer norm in front writes its operand image in the format of that layer's input site."""
        with vs.variable_scope("filter_output{}".format(self.scope_id)):
            return vs.peek_variable("kernel")

    def forward(self, inputs, **unused_params):
        # relu(dense) -> batch_norm twice (:741-760): the bias add and the ReLU of each dense layer ride in the batch norm's passes
        n1, n2 = "filter_output{}".format(self.scope_id), "ff_output{}".format(self.scope_id)
        rows = inputs.numel() // inputs.shape[-1]
        if (self.is_train and inputs.dim() == 3 and layers.use_split_gemm(inputs, rows, self.filter_size)
                and ops.ffn_mod_x3_ok(inputs, self.filter_size, self.final_size)):
            # dense -> relu -> batch_norm -> dense as ONE node: the [rows, 4F] tensor in the middle leaves the batch norm only as the
            # second GEMM's operand image, its gradient only as the first layer's gradient image (variables in the unfused order)
            w1, b1 = layers.dense_variables(n1, inputs.shape[-1], self.filter_size, True, inputs.device)
            g1, be1, mm1, mv1 = layers.bn_variables("filter_bn", self.filter_size, inputs.device)
            w2, b2 = layers.dense_variables(n2, self.filter_size, self.final_size, True, inputs.device)
            pre2 = ops.ffn_mod_x3(inputs, w1, b1, g1, be1, mm1, mv1, w2)
            return layers.batch_norm(pre2, self.is_train, "feed_output_bn", pre_bias=b2, pre_relu=True)
        pre, b1 = layers.dense(inputs, self.filter_size, True, n1, defer_bias=True)
        filter_output = layers.batch_norm(pre, self.is_train, "filter_bn", pre_bias=b1, pre_relu=True)
        pre2, b2 = layers.dense(filter_output, self.final_size, True, n2, defer_bias=True)
        return layers.batch_norm(pre2, self.is_train, "feed_output_bn", pre_bias=b2, pre_relu=True)


class TransformerEncoder(modules.BaseModule):
    """transformer_utils.py:374-413."""

    def __init__(self, feature_size, hidden_size, num_heads, attention_dropout, ff_filter_size, ff_relu_dropout,
                 is_train, scope_id):
        self.feature_size = feature_size
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.is_train = is_train
        self.scope_id = scope_id
        self.multi_head_attention = MultiHeadAttention(feature_size, hidden_size, num_heads, attention_dropout, is_train)
        self.ff_network = FeedForwardNetwork(feature_size, ff_filter_size, ff_relu_dropout, is_train, scope_id)

    def _fused_blocks(self, inputs, out_slot=None):
        """The same encoder as two block Functions (ops._AttnBlockX3, ops._FFNBlockX3): identical kernels and variables
        (created in the unfused path's order), but the three gradient sums of the shared tensors happen inside a GEMM /
        a layer-norm kernel instead of as add passes."""
        from . import variables as vs
        F_, dev = inputs.shape[-1], inputs.device
        hidden, filt = self.hidden_size, self.ff_network.filter_size
        wq, _ = layers.dense_variables("q", F_, hidden, False, dev)
        wk, _ = layers.dense_variables("k", F_, hidden, False, dev)
        wv, _ = layers.dense_variables("v", F_, hidden, False, dev)
        wo, bo = layers.dense_variables("output_transform", hidden, self.feature_size, True, dev)

        def ln_vars(scope):
            with vs.variable_scope(scope):
                beta = vs.get_variable("beta", [F_], vs.zeros_initializer(), device=dev)
                gamma = vs.get_variable("gamma", [F_], vs.ones_initializer(), device=dev)
            return gamma, beta
        g0, be0 = ln_vars("LayerNorm")
        w1, b1 = layers.dense_variables("filter_output{}".format(self.scope_id), F_, filt, True, dev)
        w2, b2 = layers.dense_variables("ff_output{}".format(self.scope_id), filt, self.feature_size, True, dev)
        g1, be1 = ln_vars("LayerNorm_1")
        g2, be2 = ln_vars("LayerNorm_2")
        depth = hidden // self.num_heads
        attention = ops.attention_block_x3(inputs, wq, wk, wv, wo, bo, g0, be0, self.num_heads, depth ** -0.5, next_kernel=w1)   # :403-407
        return ops.ffn_block_x3(attention, w1, b1, w2, b2, g1, be1, g2, be2, out=out_slot)                         # :409-411

    def fused(self, inputs):
        """Whether ``forward`` takes the block-Function path for this input (then it can write into an ops.OutputSlot)."""
        return inputs.dim() == 3 and inputs.shape[-1] == self.feature_size and self.fused_shape(inputs.shape[0], inputs.shape[1], inputs.is_cuda)

    def fused_shape(self, batch, tokens, is_cuda):
        """The same question before the input exists (the pooling op asks, to hand over a lazily normalised descriptor)."""
        from . import FLAGS
        return bool(FLAGS.fused_encoder_blocks and is_cuda and self.feature_size in ops.LN_FEATURES
                    and FLAGS.dense_precision == "bf16x3" and batch * tokens >= 1024 and self.feature_size % 8 == 0
                    and self.hidden_size % 8 == 0 and self.ff_network.filter_size % 8 == 0
                    and self.hidden_size // self.num_heads in (8, 16) and tokens <= 512)

    def forward(self, inputs, out_slot=None, **unused_params):
        if self.fused(inputs):
            return self._fused_blocks(inputs, out_slot)
        inputs = ops.materialise(inputs)       # (a lazily normalised descriptor is the block path's business only)
        if out_slot is not None:
            raise ValueError("out_slot needs the block-Function path (TransformerEncoder.fused)")
        attention, bias = self.multi_head_attention.forward(inputs, inputs, defer_bias=True)
        attention = layers.layer_norm(attention, "LayerNorm", residual=inputs, bias=bias)   # attention + inputs :405-407
        ff_output = self.ff_network.forward(attention)                                 # adds its own residual + LayerNorm_1
        return layers.layer_norm(ff_output, "LayerNorm_2", residual=attention)         # ff_output + attention :409-411


class TransformerEncoderMod(modules.BaseModule):
    """transformer_utils.py:415-457."""

    def __init__(self, feature_size, hidden_size, num_heads, attention_dropout, ff_filter_size, ff_relu_dropout,
                 is_train, scope_id, final_size):
        self.attention_dropout = attention_dropout
        self.is_train = is_train
        self.multi_head_attention = MultiHeadAttentionBN(feature_size, hidden_size, num_heads, attention_dropout, is_train)
        self.ff_network = FeedForwardNetworkMod(feature_size, ff_filter_size, ff_relu_dropout, is_train, scope_id, final_size)

    def forward(self, inputs, dropout_mask=None, dropout_rate=None, grad_join=None, **unused_params):
        """grad_join (ops.GradJoin): ``inputs`` has a second reader behind this encoder; the attention-half node accepts its gradient."""
        rate = (1.0 - self.attention_dropout) if dropout_rate is None else dropout_rate
        if (self.is_train and 0.0 < rate < 1.0 and ops.LN_DROPOUT_FUSED and inputs.is_cuda and inputs.dim() == 3
                and inputs.shape[-1] in ops.LN_FEATURES):
            # output_transform's bias add and the dropout ride in the layer norm's passes (forward: z = (a + bias) * keep / (1 - rate) +
            # inputs; backward: the gradient of a leaves the layer norm masked and scaled, with the bias gradient) when the attention
            # block hands back its raw GEMM output
            mha = self.multi_head_attention
            rows = inputs.numel() // inputs.shape[-1]
            if (ops.ATTN_BLOCK_BN and layers.use_split_gemm(inputs, rows, mha.hidden_size)
                    and ops.qkv_attention_bn_ok(inputs, mha.hidden_size, mha.num_heads)
                    and layers.use_split_gemm(inputs, rows, mha.feature_size) and mha.hidden_size % 8 == 0 and mha.feature_size % 8 == 0
                    and ops.BN_DENSE_FUSED and inputs.dtype == torch.float32):
                # the whole attention half of the encoder as ONE node (ops._AttnBlockBNX3): same kernels, same variables in the same
                # order (q, k, v kernels; logits_bn; attention_bn; output_transform; LayerNorm), and the two gradients of ``inputs`` meet
                # in the q/k/v input-gradient GEMM instead of in an add pass
                dev, F_ = inputs.device, inputs.shape[-1]
                wq, _ = layers.dense_variables("q", F_, mha.hidden_size, False, dev)
                wk, _ = layers.dense_variables("k", F_, mha.hidden_size, False, dev)
                wv, _ = layers.dense_variables("v", F_, mha.hidden_size, False, dev)
                lbn = layers.bn_variables("logits_bn", inputs.shape[1], dev)
                abn = layers.bn_variables("attention_bn", mha.hidden_size, dev)
                wo, bo = layers.dense_variables("output_transform", mha.hidden_size, mha.feature_size, True, dev)
                if dropout_mask is None:
                    dropout_mask = ops.dropout_keep_mask((*inputs.shape[:-1], mha.feature_size), 1.0 - rate, dev)
                elif dropout_mask.dtype not in (torch.bool, torch.uint8):
                    dropout_mask = dropout_mask.ne(0)
                image = bool(layers.use_split_gemm(inputs, rows, self.ff_network.filter_size)
                             and ops.ffn_mod_x3_ok(inputs, self.ff_network.filter_size, self.ff_network.final_size))
                gamma, beta = layers.layer_norm_variables("LayerNorm", mha.feature_size, dev)
                attention = ops.attention_block_bn_x3(inputs, wq, wk, wv, lbn, abn, wo, bo, gamma, beta, mha.num_heads, dropout_mask,
                                                      1.0 / (1.0 - rate), image=image, next_kernel=self.ff_network.first_kernel(),
                                                      grad_join=grad_join)
                return self.ff_network.forward(attention)
            attention, bias = self.multi_head_attention.forward(inputs, inputs, defer_bias=True)
            if bias is not None:
                if dropout_mask is None:
                    dropout_mask = ops.dropout_keep_mask(attention.shape, 1.0 - rate, attention.device)
                elif dropout_mask.dtype not in (torch.bool, torch.uint8):
                    dropout_mask = dropout_mask.ne(0)          # a KEEP mask handed in by a test / the parity step: non-zero = kept
                image = bool(layers.use_split_gemm(attention, attention.numel() // attention.shape[-1], self.ff_network.filter_size)
                             and ops.ffn_mod_x3_ok(attention, self.ff_network.filter_size, self.ff_network.final_size))
                attention = layers.layer_norm(attention, "LayerNorm", residual=inputs, bias=bias, image=image, mask=dropout_mask,
                                              mask_scale=1.0 / (1.0 - rate), next_kernel=self.ff_network.first_kernel())
                return self.ff_network.forward(attention)
        else:
            attention = self.multi_head_attention.forward(inputs, inputs)
        # tf.layers.dropout(rate = 1.0 - attention_dropout) -- drops 90 % when training (:450, App. C10)
        if self.is_train and rate > 0.0:
            if dropout_mask is None:
                # keep with probability 1 - rate, scale the kept values by 1 / (1 - rate): one fused kernel each way (drawing the mask,
                # comparing, casting, multiplying and dividing as separate passes cost five over the [B, S, F] tensor)
                attention = torch.nn.functional.dropout(attention, p=rate, training=True)
            else:
                attention = attention * dropout_mask / (1.0 - rate)
        # (the layer norm also writes the operand image of the feed-forward network's first dense layer when that network runs fused)
        image = bool(self.is_train and attention.dim() == 3
                     and layers.use_split_gemm(attention, attention.numel() // attention.shape[-1], self.ff_network.filter_size)
                     and ops.ffn_mod_x3_ok(attention, self.ff_network.filter_size, self.ff_network.final_size))
        attention = layers.layer_norm(attention, "LayerNorm", residual=inputs, image=image,
                                      next_kernel=self.ff_network.first_kernel() if image else None)         # :451-454
        return self.ff_network.forward(attention)
