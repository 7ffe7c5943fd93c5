"""MoeModel (reference: video_level_models.py:48-158) -- host-side PyTorch per the north-star."""
from __future__ import annotations

import torch

from . import FLAGS, models
from . import variables as vs


class MoeModel(models.BaseModel):
    """A softmax over a mixture of logistic models (with L2 regularization)."""

    def create_model(self, model_input, vocab_size, is_training=True, num_mixtures=None, l2_penalty=1e-8,
                     labels=None, fused_cross_entropy=False, **unused_params):
        """labels + fused_cross_entropy (the trainer sets it when its loss is CrossEntropyLoss): the mixture tail and the
        loss run as one fused kernel pair and the result carries "loss" (train.py:291-294 takes it from there)."""
        num_mixtures = num_mixtures or FLAGS.moe_num_mixtures
        l2_penalty = FLAGS.moe_l2                                     # :78 (the kwarg is ignored, App. C15)
        if FLAGS.moe_low_rank_gating != -1 or FLAGS.moe_prob_gating:
            raise NotImplementedError("low-rank / probability gating branches are off by default and not built")
        H = model_input.shape[1]
        dev = model_input.device
        store = vs.default_store()
        with vs.variable_scope("gates"):                               # slim.fully_connected, no bias :86-93
            wg = vs.get_variable("weights", [H, vocab_size * (num_mixtures + 1)], vs.glorot_uniform_initializer(), device=dev)
        with vs.variable_scope("experts"):                             # :109-114
            we = vs.get_variable("weights", [H, vocab_size * num_mixtures], vs.glorot_uniform_initializer(), device=dev)
            be = vs.get_variable("biases", [vocab_size * num_mixtures], vs.zeros_initializer(), device=dev)
        store.add_l2_regularizer(wg, l2_penalty)                            # slim.l2_regularizer :91
        store.add_l2_regularizer(we, l2_penalty)                            # :113
        gate_activations = model_input.matmul(wg)
        expert_activations = torch.addmm(be, model_input, we)
        if model_input.is_cuda and num_mixtures <= 8 and (labels is None or fused_cross_entropy):
            from . import ops
            predictions, loss = ops.moe_cross_entropy(gate_activations, expert_activations,
                                                      labels if fused_cross_entropy else None, num_mixtures)   # :116-126 (+ losses.py:41-51)
            return {"predictions": predictions, "loss": loss} if loss is not None else {"predictions": predictions}
        gating_distribution = torch.softmax(gate_activations.reshape(-1, num_mixtures + 1), dim=-1)   # :116-118
        expert_distribution = torch.sigmoid(expert_activations.reshape(-1, num_mixtures))             # :119-121
        probabilities = (gating_distribution[:, :num_mixtures] * expert_distribution).sum(dim=1)      # :123-124
        return {"predictions": probabilities.reshape(-1, vocab_size)}                                 # :125-126,158
