"""Loss functions of the packed path (mirrors touchnet/loss/)."""
from .cross_entropy import cross_entropy_loss, fused_linear_cross_entropy  # noqa: F401
