"""Context parallelism for the packed path: the sequence dimension of every [B, T, ...] buffer is split over the
`cp` group with head/tail load balancing, K/V are all-gathered per layer and each rank runs the document-masked
attention kernel on ITS query rows against the full K/V; dK/dV partial sums return by reduce-scatter.

What it replaces: `create_context_parallel_ctx` / `get_train_context` (touchnet/utils/distributed.py:292-346,
used at touchnet/bin/train.py:354-389) = torch's experimental ring attention, which patches SDPA only and so
cannot be combined with the packed document mask anywhere in the reference.  Same sharding
(2*cp chunks, rank r owns chunks r and 2cp-1-r), same default rotate method ("allgather"), same loss algebra
(the per-sentence-normalised loss is additive over sequence shards, tests/touchnet/utils/test_pack_loss.py).

MI355X notes: the all-gather of K+V (2*T*Nkv*D*2 B per layer) and the reduce-scatter of dK+dV ride RCCL over
xGMI and are independent of the query-side compute, so they can be issued on a side stream one layer ahead;
document-aware halo trimming (skip gathering chunks no local document reaches) is the next step — the kernel
already skips their tiles.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


def _backend(group) -> str:
    return dist.get_backend(group)


def _all_gather(x: torch.Tensor, group) -> torch.Tensor:
    """[...] -> [cp, ...] stacked by rank."""
    w = dist.get_world_size(group)
    out = torch.empty((w,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    if _backend(group) == "gloo":                      # CPU tests
        dist.all_gather(list(out.unbind(0)), x.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def _reduce_scatter(x: torch.Tensor, group) -> torch.Tensor:
    """[cp, ...] (rank-major) -> [...] summed over ranks, rank r keeps slice r."""
    w, r = dist.get_world_size(group), dist.get_rank(group)
    if _backend(group) == "gloo":
        x = x.contiguous()
        dist.all_reduce(x, group=group)
        return x[r].clone()
    out = torch.empty(x.shape[1:], dtype=x.dtype, device=x.device)
    dist.reduce_scatter_tensor(out, x.contiguous(), group=group)
    return out


@dataclass
class ContextParallel:
    group: object
    T: int                     # global packed length

    def __post_init__(self):
        self.cp = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        if self.T % (2 * self.cp * 128):
            raise ValueError(f"T={self.T} must be a multiple of 2*cp*128 = {2 * self.cp * 128}")
        self.Tc = self.T // (2 * self.cp)

    # ---- sharding ------------------------------------------------------------------------------------
    def chunk_owner(self, c: int):
        """global chunk c of 2*cp -> (rank, half)"""
        return (c, 0) if c < self.cp else (2 * self.cp - 1 - c, 1)

    def shard(self, x: torch.Tensor, dim: int = 1) -> torch.Tensor:
        a = x.narrow(dim, self.rank * self.Tc, self.Tc)
        b = x.narrow(dim, (2 * self.cp - 1 - self.rank) * self.Tc, self.Tc)
        return torch.cat([a, b], dim=dim).contiguous()

    def seq_shard(self):
        from touchnet_amd.functional import SeqShard
        return SeqShard(((0, self.Tc, self.rank * self.Tc),
                         (self.Tc, self.Tc, (2 * self.cp - 1 - self.rank) * self.Tc)), 2 * self.Tc)

    # ---- differentiable collectives over the sequence dim (dim 1) ------------------------------------------
    def gather_seq(self, x_local: torch.Tensor) -> torch.Tensor:
        """[B, 2*Tc, ...] local -> [B, T, ...] global; backward = reduce-scatter of the gradient."""
        return _GatherSeq.apply(x_local, self)

    def _to_global(self, stacked: torch.Tensor) -> torch.Tensor:        # [cp, B, 2Tc, ...] -> [B, T, ...]
        parts = []
        for c in range(2 * self.cp):
            r, half = self.chunk_owner(c)
            parts.append(stacked[r].narrow(1, half * self.Tc, self.Tc))
        return torch.cat(parts, dim=1)

    def _to_rank_major(self, full: torch.Tensor) -> torch.Tensor:       # [B, T, ...] -> [cp, B, 2Tc, ...]
        rows = []
        for r in range(self.cp):
            rows.append(torch.cat([full.narrow(1, r * self.Tc, self.Tc),
                                   full.narrow(1, (2 * self.cp - 1 - r) * self.Tc, self.Tc)], dim=1))
        return torch.stack(rows, dim=0)


class _GatherSeq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, cp: ContextParallel):
        ctx.cp = cp
        return cp._to_global(_all_gather(x_local, cp.group))

    @staticmethod
    def backward(ctx, g_full):
        cp = ctx.cp
        return _reduce_scatter(cp._to_rank_major(g_full), cp.group), None
