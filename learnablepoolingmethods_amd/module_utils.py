"""Regularisers (reference: module_utils.py:55-90)."""
import numbers

import torch

from . import layers


def orthogonal_regularizer(scale, scope=None):
    """Returns ``orthogonal_sum(weights) = scale * sum |W^T W - I|`` with W = l2_normalize(weights, axis=1), or a function
    returning None when scale == 0 (module_utils.py:55-90; the tf.Print of the unscaled value is not reproduced)."""
    if isinstance(scale, numbers.Integral):
        raise ValueError("scale cannot be an integer: %s" % (scale,))
    if isinstance(scale, numbers.Real):
        if scale < 0.0:
            raise ValueError("Setting a scale less than 0 on a regularizer: %g." % scale)
        if scale == 0.0:
            return lambda _: None

    def orthogonal_sum(weights):
        w = layers.l2_normalize(weights, 1)
        det_reg = w.t().matmul(w) - torch.eye(w.shape[1], dtype=w.dtype, device=w.device)
        return scale * det_reg.abs().sum()
    return orthogonal_sum
