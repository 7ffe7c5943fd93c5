"""Frame sampling (reference: model_utils.py:26-122)."""
import torch

from . import ops


def SampleUniformFrames(model_input, num_frames, num_samples):
    """Deterministic uniform sampling: [B, max_frames, F] -> [B, num_samples, F] (HIP gather kernel)."""
    B, _, F = model_input.shape
    return ops.frame_sample_bn(model_input, num_frames.reshape(-1), num_samples).reshape(B, num_samples, F)


def _gather_frames(model_input, frame_index):
    batch_index = torch.arange(model_input.shape[0], device=model_input.device).unsqueeze(1).expand_as(frame_index)
    return model_input[batch_index, frame_index]


def SampleRandomSequence(model_input, num_frames, num_samples, uniform=None):
    """model_utils.py:26-57: a contiguous run of num_samples frames from a random start (indices clamped to num_frames - 1).
    ``uniform`` [B, 1] ~ U[0,1) replaces tf.random_uniform (drawn here when None) so that a run can be reproduced."""
    B = model_input.shape[0]
    nf = num_frames.reshape(-1, 1).to(device=model_input.device, dtype=torch.float32)
    if uniform is None:
        uniform = torch.rand((B, 1), device=model_input.device)
    max_start = torch.clamp(nf - num_samples, min=0.0)
    start = (uniform.to(model_input.device, torch.float32).reshape(-1, 1) * (max_start + 1.0)).to(torch.int32)
    offset = torch.arange(num_samples, device=model_input.device, dtype=torch.int32).unsqueeze(0)
    frame_index = torch.minimum(start + offset, (nf - 1).to(torch.int32)).long()
    return _gather_frames(model_input, frame_index)


def SampleRandomFrames(model_input, num_frames, num_samples, uniform=None):
    """model_utils.py:60-78: num_samples frames drawn independently, idx = int32(u * num_frames); ``uniform`` [B, S] as above."""
    B = model_input.shape[0]
    nf = num_frames.reshape(-1, 1).to(device=model_input.device, dtype=torch.float32)
    if uniform is None:
        uniform = torch.rand((B, num_samples), device=model_input.device)
    frame_index = (uniform.to(model_input.device, torch.float32) * nf).to(torch.int32).long()
    return _gather_frames(model_input, frame_index)
