from .fused_sgd import FusedSGD  # noqa: F401
