"""learnablepoolingmethods_amd -- MI355X-native NetVLAD / attention-pooling training path.

Drop-in for the NetVladV1 / NetVladV2 hot path of pomonam/LearnablePoolingMethods
(frame_level_models.py / video_pooling_modules.py / transformer_utils.py / video_level_models.py /
train.py): the same registry, ``create_model`` / ``forward`` conventions and variable names, with the
hot ops running as hand-written gfx950 HIP kernels behind a C ABI (include/lpm_hip.h).
"""
from . import flags as _flags

FLAGS = _flags.FLAGS
__version__ = "0.1.0"
