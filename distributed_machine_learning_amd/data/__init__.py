from .datasets import ArrayDataset, synthetic_cifar, CifarBinary, build_datasets  # noqa: F401
from .sampler import ShardedSampler, BatchLoader  # noqa: F401
from .transforms import BatchTransform  # noqa: F401
