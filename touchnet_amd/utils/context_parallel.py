"""Context parallelism for the packed path: the sequence dimension of every [B, T, ...] buffer is split over the
`cp` group with head/tail load balancing, K/V are all-gathered per layer and each rank runs the document-masked
attention kernel on ITS query rows against the full K/V; dK/dV partial sums return by reduce-scatter.

What it replaces: `create_context_parallel_ctx` / `get_train_context` (touchnet/utils/distributed.py:292-346,
used at touchnet/bin/train.py:354-389) = torch's experimental ring attention, which patches SDPA only and so
cannot be combined with the packed document mask anywhere in the reference.  Same sharding
(2*cp chunks, rank r owns chunks r and 2cp-1-r), same default rotate method ("allgather"), same loss algebra
(the per-sentence-normalised loss is additive over sequence shards, tests/touchnet/utils/test_pack_loss.py).

MI355X notes: xGMI is point-to-point, so the exchange does not have to be a collective.  With `set_documents`
(the global document ids of the batch: tiny, every rank has them) each rank knows which of the 2*cp chunks ANY rank
needs — a chunk is needed by rank r iff it is one of r's own, or it precedes one of r's chunks and shares a document
with it — and `gather_seq` becomes a HALO EXCHANGE: batched isend/irecv of exactly those chunks (RCCL send/recv
over the direct links), zeros elsewhere (the kernels never let a query see them: their tiles are skipped or fully
masked); the backward sends the dK/dV partial sums back over the same pairs.  On packed batches of documents much
shorter than a chunk a rank receives 1-2 neighbour chunks per layer instead of 2*cp - 2.  Without `set_documents`
the all-gather / reduce-scatter pair (the reference's "allgather" rotate method) is used.
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


def halo_need(ids, cp: int):
    """ids int [B, T] (0 = pad) -> bool [cp, 2cp]: rank r needs chunk c iff c is one of r's chunks (r and 2cp-1-r), or
    c < lc for a chunk lc of r (causality) and some batch row has a positive document id present in both chunks.
    Exact at chunk granularity: no (query, key) pair the document mask allows is ever left out."""
    import numpy as np
    C = 2 * cp
    ch = np.asarray(ids).reshape(ids.shape[0], C, -1)
    present = [[np.unique(ch[b, c][ch[b, c] > 0]) for c in range(C)] for b in range(ch.shape[0])]
    need = np.zeros((cp, C), dtype=bool)
    for r in range(cp):
        for lc in (r, C - 1 - r):
            need[r, lc] = True
            for c in range(lc):
                if not need[r, c]:
                    need[r, c] = any(np.intersect1d(present[b][lc], present[b][c], assume_unique=True).size
                                     for b in range(ch.shape[0]))
    return need


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
        self.need = None                 # bool [cp, 2cp] after set_documents(): rank r needs chunk c
        self.halo_bytes = 0              # bytes this rank received through the halo exchange (diagnostics / tests)

    # ---- document-aware halo ---------------------------------------------------------------------------
    def set_documents(self, doc_ids: torch.Tensor) -> None:
        """doc_ids int [B, T] (0 = pad), identical on every rank: fixes which chunks each rank exchanges (halo_need)."""
        self.need = halo_need(doc_ids.detach().to("cpu").numpy(), self.cp)

    def my_chunks(self, r=None):
        r = self.rank if r is None else r
        return (r, 2 * self.cp - 1 - r)

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
        """[B, 2*Tc, ...] local -> [B, T, ...] global; backward = reduce-scatter of the gradient.  After
        `set_documents`: halo exchange of the needed chunks only (zeros elsewhere)."""
        if self.need is not None:
            return _HaloExchange.apply(x_local, self)
        return _GatherSeq.apply(x_local, self)

    def _exchange(self, pieces_for, recv_from, like: torch.Tensor):
        """Batched point-to-point: `pieces_for[p]` = tensor to send to group rank p (or None), `recv_from[p]` = number
        of chunks expected from p.  Returns {p: tensor [B, k*Tc, ...]}."""
        ops, got = [], {}
        for p in range(self.cp):
            if p == self.rank:
                continue
            peer = dist.get_global_rank(self.group, p)
            if pieces_for.get(p) is not None:
                ops.append(dist.P2POp(dist.isend, pieces_for[p], peer, group=self.group))
            k = recv_from.get(p, 0)
            if k:
                buf = torch.empty((like.shape[0], k * self.Tc) + tuple(like.shape[2:]), dtype=like.dtype,
                                  device=like.device)
                got[p] = buf
                self.halo_bytes += buf.numel() * buf.element_size()
                ops.append(dist.P2POp(dist.irecv, buf, peer, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return got

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


class _HaloExchange(torch.autograd.Function):
    """gather_seq restricted to the chunks `cp.need` says a rank can see (see the module docstring)."""

    @staticmethod
    def forward(ctx, x_local, cp: ContextParallel):
        ctx.cp, Tc, me = cp, cp.Tc, cp.rank
        full = x_local.new_zeros((x_local.shape[0], cp.T) + tuple(x_local.shape[2:]))
        mine = cp.my_chunks()
        for h, c in enumerate(mine):
            full.narrow(1, c * Tc, Tc).copy_(x_local.narrow(1, h * Tc, Tc))
        send, recv = {}, {}
        for p in range(cp.cp):
            if p == me:
                continue
            halves = [h for h, c in enumerate(mine) if cp.need[p, c]]
            if halves:
                send[p] = torch.cat([x_local.narrow(1, h * Tc, Tc) for h in halves], dim=1).contiguous()
            recv[p] = sum(bool(cp.need[me, c]) for c in cp.my_chunks(p))
        got = cp._exchange(send, recv, x_local)
        for p, buf in got.items():
            cs = [c for c in cp.my_chunks(p) if cp.need[me, c]]
            for i, c in enumerate(cs):
                full.narrow(1, c * Tc, Tc).copy_(buf.narrow(1, i * Tc, Tc))
        return full

    @staticmethod
    def backward(ctx, g_full):
        cp, Tc, me = ctx.cp, ctx.cp.Tc, ctx.cp.rank
        mine = cp.my_chunks()
        g_local = torch.cat([g_full.narrow(1, c * Tc, Tc) for c in mine], dim=1).contiguous()
        send, recv = {}, {}
        for p in range(cp.cp):
            if p == me:
                continue
            cs = [c for c in cp.my_chunks(p) if cp.need[me, c]]          # chunks of p this rank used: grads go back
            if cs:
                send[p] = torch.cat([g_full.narrow(1, c * Tc, Tc) for c in cs], dim=1).contiguous()
            recv[p] = sum(bool(cp.need[p, c]) for c in mine)              # p used these chunks of mine
        got = cp._exchange(send, recv, g_local)
        for p, buf in got.items():
            halves = [h for h, c in enumerate(mine) if cp.need[p, c]]
            for i, h in enumerate(halves):
                g_local.narrow(1, h * Tc, Tc).add_(buf.narrow(1, i * Tc, Tc))
        return g_local, None
