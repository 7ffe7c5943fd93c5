"""MoeModel (reference: video_level_models.py:48-158) -- host-side PyTorch per the north-star."""
from __future__ import annotations

import math

import torch

from . import FLAGS, models
from . import variables as vs


def _linear(x, W, b=None):
    """x W (+ b); on the GPU through ops.linear_direct (the weight gradient may go straight into the trainer's gradient arena)."""
    if x.is_cuda and x.dtype == torch.float32 and W.dtype == torch.float32 and x.dim() == 2:
        from . import ops
        return ops.linear_direct(x, W, b)
    return torch.addmm(b, x, W) if b is not None else x.matmul(W)


class MoeModel(models.BaseModel):
    """A softmax over a mixture of logistic models (with L2 regularization)."""

    def create_model(self, model_input, vocab_size, is_training=True, num_mixtures=None, l2_penalty=1e-8,
                     labels=None, fused_cross_entropy=False, **unused_params):
        """labels + fused_cross_entropy (the trainer sets it when its loss is CrossEntropyLoss): the mixture tail and the
        loss run as one fused kernel pair and the result carries "loss" (train.py:291-294 takes it from there)."""
        num_mixtures = num_mixtures or FLAGS.moe_num_mixtures
        low_rank_gating = FLAGS.moe_low_rank_gating                    # :77
        l2_penalty = FLAGS.moe_l2                                     # :78 (the kwarg is ignored, App. C15)
        gating_probabilities = FLAGS.moe_prob_gating                   # :79
        gating_input = FLAGS.moe_prob_gating_input                     # :80
        H = model_input.shape[1]
        dev = model_input.device
        store = vs.default_store()
        if low_rank_gating == -1:
            with vs.variable_scope("gates"):                           # slim.fully_connected, no bias :86-93
                wg = vs.get_variable("weights", [H, vocab_size * (num_mixtures + 1)], vs.glorot_uniform_initializer(), device=dev)
            store.add_l2_regularizer(wg, l2_penalty)                   # slim.l2_regularizer :91
            gate_activations = _linear(model_input, wg)
        else:                                                          # two bias-free layers through a low_rank_gating bottleneck :94-108
            with vs.variable_scope("gates1"):
                wg1 = vs.get_variable("weights", [H, low_rank_gating], vs.glorot_uniform_initializer(), device=dev)
            with vs.variable_scope("gates2"):
                wg2 = vs.get_variable("weights", [low_rank_gating, vocab_size * (num_mixtures + 1)], vs.glorot_uniform_initializer(), device=dev)
            store.add_l2_regularizer(wg1, l2_penalty)                  # :99
            store.add_l2_regularizer(wg2, l2_penalty)                  # :106
            gate_activations = model_input.matmul(wg1).matmul(wg2)
        with vs.variable_scope("experts"):                             # :109-114
            we = vs.get_variable("weights", [H, vocab_size * num_mixtures], vs.glorot_uniform_initializer(), device=dev)
            be = vs.get_variable("biases", [vocab_size * num_mixtures], vs.zeros_initializer(), device=dev)
        store.add_l2_regularizer(we, l2_penalty)                       # :113
        expert_activations = _linear(model_input, we, be)
        fuse_loss = fused_cross_entropy and not gating_probabilities   # (probability gating changes the predictions behind the mixture)
        if model_input.is_cuda and num_mixtures <= 8 and (labels is None or fused_cross_entropy):
            from . import ops
            probabilities, loss = ops.moe_cross_entropy(gate_activations, expert_activations,
                                                        labels if fuse_loss else None, num_mixtures)   # :116-126 (+ losses.py:41-51)
            if not gating_probabilities:
                return {"predictions": probabilities, "loss": loss} if loss is not None else {"predictions": probabilities}
        else:
            gating_distribution = torch.softmax(gate_activations.reshape(-1, num_mixtures + 1), dim=-1)   # :116-118
            expert_distribution = torch.sigmoid(expert_activations.reshape(-1, num_mixtures))             # :119-121
            probabilities = (gating_distribution[:, :num_mixtures] * expert_distribution).sum(dim=1)      # :123-124
            probabilities = probabilities.reshape(-1, vocab_size)                                         # :125-126
        if gating_probabilities:                                       # :128-156
            from . import layers
            rows = vocab_size if gating_input == "prob" else H
            gating_weights = vs.get_variable("gating_prob_weights", [rows, vocab_size],
                                             vs.random_normal_initializer(1 / math.sqrt(vocab_size)), device=dev)   # :130-140
            gates = (probabilities if gating_input == "prob" else model_input).matmul(gating_weights)
            if FLAGS.gating_remove_diag:                               # :144-147 (tf.matrix_diag_part: the main diagonal)
                if rows != vocab_size:
                    raise ValueError("gating_remove_diag with moe_prob_gating_input != 'prob': the diagonal of a "
                                     f"[{rows}, {vocab_size}] matrix does not broadcast against the probabilities (as in the reference)")
                gates = gates - torch.diagonal(gating_weights) * probabilities
            gates = layers.batch_norm(gates, is_training, "gating_prob_bn")                                # :149-154
            probabilities = probabilities * torch.sigmoid(gates)                                           # :156-158
        return {"predictions": probabilities}                                                             # :158
