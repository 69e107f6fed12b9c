"""Oracle op namespace with the SAME signatures as touchnet_amd.functional (TEST INFRASTRUCTURE).

Lets the CPU test-suite run the product's module wiring / packers / sharding logic end to end
(`touchnet_amd.models.backend.use_ops(oracle.ops)`), and serves as the checker for the GPU path.
Every op is the eager restatement from oracle/nn.py / oracle/loss.py; layouts follow the product
([B, T, heads, D] attention tensors, [B*T, D/2] rope tables).
"""
from types import SimpleNamespace

import torch

from . import loss as _loss
from . import nn as _nn


def rms_norm(x, weight, eps, residual=None):
    if residual is None:
        return _nn.rms_norm(x, weight.to(x.dtype), eps)
    h = x + residual
    return _nn.rms_norm(h, weight.to(x.dtype), eps), h


def layer_norm(x, weight, bias, eps=1e-5, residual=None):
    if residual is None:
        return _nn.layer_norm(x, weight.to(x.dtype), bias.to(x.dtype), eps)
    h = x + residual
    return _nn.layer_norm(h, weight.to(x.dtype), bias.to(x.dtype), eps), h


swiglu = _nn.swiglu


def norm_source(h, norm_weight, eps):
    """mirror of touchnet_amd.functional.norm_source (a memory hint for the HIP path's autograd nodes: no arithmetic)"""
    return (h, norm_weight, float(eps))


def swiglu_mlp(x, w_gate, w_up, w_down, norm_src=None):
    """mirror of touchnet_amd.functional.swiglu_mlp (modeling_llama.py:174-176)"""
    lin = torch.nn.functional.linear
    return lin(swiglu(lin(x, w_gate), lin(x, w_up)), w_down)


def pcm16_to_float(pcm):
    """datapipe.py:164"""
    return pcm.to(torch.float32) / 32768.0


def linear_group(x, layers, wgrad="tn", dgrad_tn=True, norm_src=None, rope=None, rope_grad_in_attention=False):
    """mirror of touchnet_amd.functional.linear_group: plain nn.Linear math per layer; `rope = (cos, sin, head_dim, which)`:
    the outputs `which` come back rotated (apply_rope below on [.., heads, head_dim]; autograd rotates their gradients
    back whoever consumes them, so `rope_grad_in_attention` / packed_attention's `rope_grad` change nothing here)"""
    outs = [torch.nn.functional.linear(x, w, b) for w, b in layers]
    if rope is not None:
        cos, sin, D, which = rope
        for i in which:
            y = outs[i]
            y4 = y.reshape(y.shape[0], y.shape[1], -1, D) if y.dim() == 3 else y.reshape(1, y.shape[0], -1, D)
            outs[i] = apply_rope(y4, y4[:, :, :0], cos, sin)[0].reshape(y.shape)
    return outs


gelu = torch.nn.functional.gelu


def gelu_mlp(x, w1, b1, w2, b2):
    """mirror of touchnet_amd.functional.gelu_mlp: WhisperEncoderLayer's fc2(gelu(fc1(x))), exact-erf GELU"""
    lin = torch.nn.functional.linear
    return lin(gelu(lin(x, w1, b1)), w2, b2)


def rope_inv_freq(head_dim, theta, scaling=None, device=None):
    inv = _nn.rope_inv_freq(head_dim, theta, scaling)
    return inv.to(device) if device is not None else inv


def rope_tables(position_ids, inv_freq, dtype, attention_scaling=1.0):
    cos, sin = _nn.rope_cos_sin(position_ids, inv_freq, dtype)
    half = inv_freq.numel()
    return cos[..., :half].reshape(-1, half), sin[..., :half].reshape(-1, half)


def apply_rope(q, k, cos, sin):
    B, T = q.shape[:2]
    c = torch.cat([cos, cos], -1).view(B, T, -1)
    s = torch.cat([sin, sin], -1).view(B, T, -1)
    qo, ko = _nn.apply_rope(q.transpose(1, 2), k.transpose(1, 2), c, s)
    return qo.transpose(1, 2), ko.transpose(1, 2)


def build_packed_mask(doc_ids):
    return SimpleNamespace(doc=doc_ids, allow=_nn.doc_causal_allow(doc_ids), B=doc_ids.shape[0], T=doc_ids.shape[1])


def causal_mask(B, T, device):
    return build_packed_mask(torch.ones(B, T, dtype=torch.int64, device=device))


def packed_attention(q, k, v, mask, scale=None, rope_grad=None):
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    return _nn.attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask.allow, scale)


def bidirectional_attention(q, k, v, mask, scale=None):
    """every key of the query's own document, before and after it (transformers' WhisperEncoder self-attention as used at
    touchnet/models/kimi_audio/modeling_kimi_audio.py:337-339, 942-947); pad rows give 0"""
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    doc = mask.doc
    allow = (doc[:, :, None] > 0) & (doc[:, :, None] == doc[:, None, :])
    return _nn.attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), allow, scale)


def _num_sentence_dev(num_sentence, device):
    if isinstance(num_sentence, torch.Tensor):
        return num_sentence.to(device=device, dtype=torch.float32).reshape(1)
    return torch.tensor([float(num_sentence)], dtype=torch.float32, device=device)


def packed_cross_entropy(pred, labels, sentence_lens, num_sentence, ignore_index=-100, inplace_grad=False):
    ns = _num_sentence_dev(num_sentence, pred.device)[0]
    ps, pt = _loss.cross_entropy_loss(pred, labels, sentence_lens, ns, ignore_index)
    acc = _loss.accuracy(pred.detach(), labels, ignore_index)
    nvalid = (labels != ignore_index).sum().float()
    return ps, torch.stack([ps.detach(), pt.detach(), acc.float(), nvalid])


class _GatherVocab(torch.autograd.Function):
    """all-gather of vocabulary-sharded logits along the last dim; the loss behind it is computed identically on every
    rank, so the gradient of the local logits is the local slice of d(logits) (nothing to sum)"""

    @staticmethod
    def forward(ctx, x, group, rank, size):
        import torch.distributed as dist
        ctx.rank, ctx.v = rank, x.shape[-1]
        parts = [torch.empty_like(x) for _ in range(size)]
        dist.all_gather(parts, x.contiguous(), group=group)
        return torch.cat(parts, dim=-1)

    @staticmethod
    def backward(ctx, g):
        return g[..., ctx.rank * ctx.v:(ctx.rank + 1) * ctx.v].contiguous(), None, None, None


def fused_linear_cross_entropy(hidden, weight, labels, sentence_lens, num_sentence, ignore_index=-100,
                               chunk_tokens=16384, compact=False, tp=None):
    """`tp = (group, rank, size)`: `weight` is this rank's vocabulary shard (loss parallel,
    touchnet/loss/cross_entropy.py:29-33 / parallelize_llama.py:177-186): the full logits are gathered and the plain
    reference loss is evaluated on them; d(hidden) comes out as the partial sum over the local shard, like in the product."""
    logits = torch.nn.functional.linear(hidden, weight)
    if tp is not None:
        logits = _GatherVocab.apply(logits, *tp)
    return packed_cross_entropy(logits, labels, sentence_lens, num_sentence, ignore_index)


def packed_attention_sharded(q_local, k_full, v_full, mask, shard, scale=None):
    """Local query rows (global positions per `shard.segs`) against the full K/V — the CP building block."""
    scale = q_local.shape[-1] ** -0.5 if scale is None else scale
    pos = torch.cat([torch.arange(off, off + rows) for (_, rows, off) in shard.segs]).to(q_local.device)
    allow = mask.allow[:, pos, :]                                  # [B, R, T]
    B, R, Nh, D = q_local.shape
    g = Nh // k_full.shape[2]
    k = k_full.transpose(1, 2).repeat_interleave(g, dim=1)
    v = v_full.transpose(1, 2).repeat_interleave(g, dim=1)
    s = torch.matmul(q_local.transpose(1, 2), k.transpose(2, 3)) * scale
    s = s.masked_fill(~allow[:, None], torch.finfo(s.dtype).min)
    p = torch.softmax(s, dim=-1, dtype=torch.float32).to(q_local.dtype)
    p = p * allow.any(-1)[:, None, :, None].to(p.dtype)
    return torch.matmul(p, v).transpose(1, 2).contiguous()


def packed_attention_sharded_split(q_local, k_full, v_full, mask, shard, chunk_len, own_chunks, remote_chunks, wait=None,
                                   scale=None):
    """mirror of touchnet_amd.functional.packed_attention_sharded_split: the softmax over the rank's OWN chunks, then
    (behind `wait()`) over the received chunks, merged by their log-sum-exp — the formula of torch's ring attention
    (torch/distributed/tensor/experimental/_context_parallel/_attention.py:182-183, entered from
    touchnet/utils/distributed.py:292-315).  Only the chunks a part covers are ever indexed, so a test may leave the
    remote chunks poisoned until `wait()`."""
    scale = q_local.shape[-1] ** -0.5 if scale is None else scale
    pos = torch.cat([torch.arange(off, off + rows) for (_, rows, off) in shard.segs]).to(q_local.device)
    B, R, Nh, D = q_local.shape
    g = Nh // k_full.shape[2]
    qh = q_local.transpose(1, 2).float()                                           # [B, Nh, R, D]

    def part(chunks):
        if not chunks:
            return None
        cols = torch.cat([torch.arange(c * chunk_len, (c + 1) * chunk_len) for c in chunks]).to(q_local.device)
        allow = mask.allow[:, pos][:, :, cols]                                     # [B, R, n]
        k = k_full[:, cols].transpose(1, 2).repeat_interleave(g, dim=1).float()
        v = v_full[:, cols].transpose(1, 2).repeat_interleave(g, dim=1).float()
        s = torch.matmul(qh, k.transpose(2, 3)) * scale
        s = s.masked_fill(~allow[:, None], float("-inf"))
        lse = torch.logsumexp(s, dim=-1)                                           # -inf where the part is empty
        p = torch.exp(s - torch.where(torch.isfinite(lse), lse, torch.zeros_like(lse))[..., None])
        p = torch.where(allow[:, None], p, torch.zeros_like(p))
        return torch.matmul(p, v), lse                                             # [B, Nh, R, D], [B, Nh, R]

    a = part(list(own_chunks))
    if wait is not None:
        wait()
    b = part(list(remote_chunks))
    if b is None:
        o = a[0]
    else:
        lse = torch.logaddexp(a[1], b[1])
        wa = torch.where(torch.isfinite(a[1]), torch.exp(a[1] - lse), torch.zeros_like(lse))
        wb = torch.where(torch.isfinite(b[1]), torch.exp(b[1] - lse), torch.zeros_like(lse))
        o = wa[..., None] * a[0] + wb[..., None] * b[0]
        o = torch.where(torch.isfinite(lse)[..., None], o, torch.zeros_like(o))
    return o.to(q_local.dtype).transpose(1, 2).contiguous()


# ---- frontend (same call signatures as touchnet_amd.functional; numpy restatements of oracle/frontend.py) ----------
def kaldi_fbank(wav, num_mel_bins=80):
    from . import frontend as _fe
    return torch.from_numpy(_fe.kaldi_fbank(wav.detach().cpu().numpy().reshape(-1), num_mel_bins=num_mel_bins)).float()


def log_mel_spectrogram(wav, num_mel_bins=128, padding=0):
    from . import frontend as _fe
    return torch.from_numpy(_fe.log_mel_spectrogram(wav.detach().cpu().numpy().reshape(-1), n_mels=num_mel_bins,
                                                    padding=padding)).float()


def speed_perturb(wave, speed):
    from . import frontend as _fe
    return torch.from_numpy(_fe.speed_perturb(wave.detach().cpu().numpy(), speed))


def feat_augment(feat, t_masks=(), f_masks=(), subs=(), out_rows=None):
    """the product op's contract (touchnet_amd.functional.feat_augment) on the CPU: stripes, substitutions, trim"""
    x = feat.detach().cpu().float()
    y = x.clone()
    for a, b, pos in subs:
        y[a:b] = x[a - pos:b - pos]
    keep = torch.ones(x.shape[0], dtype=torch.bool)
    for a, b in t_masks:
        keep[a:b] = False
    src = torch.arange(x.shape[0])
    for a, b, pos in subs:
        src[a:b] = torch.arange(a - pos, b - pos)
    y[~keep[src]] = 0
    for a, b in f_masks:
        y[:, a:b] = 0
    return y[:x.shape[0] if out_rows is None else out_rows].clone()


def audiofeat_stack(feat, stack, stride, normalize=True):
    from . import frontend as _fe
    return torch.from_numpy(_fe.audiofeat_stack(feat.detach().cpu().numpy(), stack, stride, normalize)).float()

