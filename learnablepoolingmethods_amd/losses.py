"""Losses (reference: losses.py:21-51)."""
import torch


class BaseLoss(object):
    def calculate_loss(self, unused_predictions, unused_labels, **unused_params):
        raise NotImplementedError()


class CrossEntropyLoss(BaseLoss):
    """losses.py:41-51: epsilon = 10e-6, sum over classes, mean over the batch."""

    def calculate_loss(self, predictions, labels, **unused_params):
        epsilon = 10e-6
        float_labels = labels.to(predictions.dtype)
        cross_entropy_loss = float_labels * torch.log(predictions + epsilon) + \
            (1 - float_labels) * torch.log(1 - predictions + epsilon)
        return (-cross_entropy_loss).sum(dim=1).mean()
