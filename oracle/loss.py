"""Oracle: packed, per-sentence-normalised cross-entropy + accuracy (TEST INFRASTRUCTURE).

Restates
  * ``_cross_entropy_loss``   touchnet/loss/__init__.py:7-23
  * ``cross_entropy_loss``    touchnet/loss/cross_entropy.py:12-50
  * ``accuracy``              touchnet/utils/metrics.py:26-50
in eager fp32 PyTorch (no torch.compile, no DTensor branch).
"""
import torch
import torch.nn.functional as F


def token_nll(pred, labels, ignore_index=-100):
    """loss/__init__.py:20-23 — upcast to fp32, CE(reduction='none'); 0 on ignored."""
    return F.cross_entropy(pred.flatten(0, 1).float(), labels.flatten(0, 1),
                           reduction="none", ignore_index=ignore_index)


def cross_entropy_loss(pred, labels, sentence_lens, num_sentence, ignore_index=-100):
    """loss/cross_entropy.py:34-50.

    Returns ``(loss_per_sample, loss_per_token)``:
      loss_per_token  = sum(nll) / n_valid       (0 if sum <= 1e-6 or no valid label)
      loss_per_sample = sum_b sum_t nll[b,t] / sentence_lens[b,t]  / num_sentence
    """
    bsz = pred.size(0)
    n_valid = int((labels != ignore_index).sum())
    nll = token_nll(pred, labels, ignore_index)
    tot = nll.sum()
    if tot > 1e-6 and n_valid > 0:
        per_token = tot / n_valid
    else:
        per_token = torch.zeros_like(tot)
    per_sample = (nll.reshape(bsz, -1) / sentence_lens).sum(dim=-1).sum() / num_sentence
    return per_sample, per_token


def accuracy(pred, labels, ignore_index=-100):
    """utils/metrics.py:41-50 — argmax (first index on ties) vs labels on valid slots."""
    hit = pred.argmax(dim=-1)
    mask = labels != ignore_index
    num = (hit[mask] == labels[mask]).sum()
    den = mask.sum()
    if den > 0:
        return (num / den).detach()
    return torch.zeros_like(num).detach()
