"""Frame sampling (reference: model_utils.py:101-122)."""
from . import ops


def SampleUniformFrames(model_input, num_frames, num_samples):
    """Deterministic uniform sampling: [B, max_frames, F] -> [B, num_samples, F] (HIP gather kernel)."""
    B, _, F = model_input.shape
    return ops.frame_sample_bn(model_input, num_frames.reshape(-1), num_samples).reshape(B, num_samples, F)
