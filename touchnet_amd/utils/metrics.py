"""`acc_fn` of the TrainSpec (touchnet/utils/metrics.py:26-50) and the MFU bookkeeping constants the
reference lacks for AMD (its peak table has no MI-series entry and falls back to A100,
touchnet/utils/metrics.py:67-100)."""
import torch

from touchnet_amd.loss.cross_entropy import cached_accuracy
from touchnet_amd.models.backend import ops

MI355X_BF16_DENSE_PEAK = 2.5e15      # FLOP/s, dense (no 2:1 sparsity) — MI355X_MICROARCH.md
MI355X_HBM_PEAK = 8.0e12             # B/s spec


def accuracy(pred: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """argmax(pred) == labels over non-ignored positions (first index on ties), 0-d tensor."""
    acc = cached_accuracy(pred, labels)
    if acc is not None:
        return acc.detach()
    ones = torch.ones_like(labels)
    _, stats = ops().packed_cross_entropy(pred.detach(), labels, ones, 1, ignore_index)
    return stats[2].detach()


def get_peak_flops(device_name: str = "") -> float:
    return MI355X_BF16_DENSE_PEAK
