"""Gradient combination / clipping and feature dequantisation (reference: utils.py:28-43, 170-213)."""
from __future__ import annotations

from typing import Dict, List

import torch


def Dequantize(feat_vector, max_quantized_value=2, min_quantized_value=-2):
    """utils.py:28-43."""
    assert max_quantized_value > min_quantized_value
    quantized_range = max_quantized_value - min_quantized_value
    scalar = quantized_range / 255.0
    bias = (quantized_range / 512.0) + min_quantized_value
    return feat_vector * scalar + bias


def combine_gradients(tower_grads: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """SUM (not mean) of per-tower gradients, utils.py:192-213.  In the data-parallel build the towers
    are ranks and this sum is the RCCL all-reduce in train.GradientSynchronizer; this in-process form is
    what the single-process multi-tower equivalence tests use."""
    return {n: torch.stack([tg[n] for tg in tower_grads], 0).sum(0) for n in tower_grads[0]}


def clip_gradient_norms(gradients: Dict[str, torch.Tensor], max_norm: float) -> Dict[str, torch.Tensor]:
    """Per-variable tf.clip_by_norm (utils.py:170-189): g * max_norm / max(||g||, max_norm)."""
    out = {}
    for n, g in gradients.items():
        nrm = torch.sqrt((g * g).sum())
        out[n] = g * (max_norm / torch.clamp(nrm, min=max_norm))
    return out
