"""MI355X-native data-parallel CNN training framework.

Python shell (CLI, data, loops, checkpoints) over a native C++/HIP runtime:
hand-written CDNA4 kernels (``csrc/kernels``), a static-plan training engine
(``csrc/engine``) and an RCCL communicator with a bucket reducer (``csrc/comm``).
"""
from .config import TrainConfig, part_defaults  # noqa: F401
from .models.vgg import VGG, VGG11, VGG13, VGG16, VGG19  # noqa: F401

__version__ = "0.1.0"
