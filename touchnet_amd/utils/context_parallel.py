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

Overlap (`exchange_kv`): K and V of a layer travel in ONE batch of isend/irecv that is only ISSUED when the
projections are done — the transfers then run on the communication stream of the process group (RCCL's own HIP
stream: `batch_isend_irecv` returns at once) while the compute stream goes on with the query path (q projection,
RoPE), and the compute stream waits for them right in front of the attention kernel.  The backward mirrors it: the
partial dK/dV of the remote chunks are sent back as soon as the attention backward has produced them, the query-path
backward GEMMs run meanwhile, and the owner adds the returned partial sums just before the k/v projection backward.
Two autograd nodes per layer make that order happen (the engine runs ready nodes in reverse creation order):
`_HaloStart` (issue forward / finish backward) is created BEFORE the query path, `_HaloFinish` (finish forward / issue
backward) after it.  Chunks are received straight into the global K/V buffers when a chunk is contiguous there
(B = 1: config D), which are left UNINITIALISED elsewhere — the attention kernels only touch tiles whose id range can
meet the queries', and the halo is computed with the same (conservative) range test, so every tile they touch has
been received.  `set_documents` derives the halo from chunk-level id ranges ON THE DEVICE and reads back only the
[cp, 2cp] table, asynchronously (pinned copy + event, waited at the first exchange of the step).
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


def halo_need_ranges(doc_ids: torch.Tensor, cp: int) -> torch.Tensor:
    """Device-side form of `halo_need` from per-chunk id RANGES: rank r needs chunk c < lc iff the positive-id ranges
    of c and lc overlap in some batch row (the test the attention kernels use per 64-tile, at chunk granularity).
    Equal to `halo_need` when ids are non-decreasing along a row (every packer here), a superset otherwise.
    doc_ids int [B, T] -> bool [cp, 2cp] on the same device; no host synchronisation."""
    C = 2 * cp
    B = doc_ids.shape[0]
    ch = doc_ids.reshape(B, C, -1).to(torch.int64)
    big = torch.iinfo(torch.int64).max
    mx = ch.amax(dim=2)                                                 # [B, C] (0 = chunk is all pad)
    mn = torch.where(ch > 0, ch, torch.full_like(ch, big)).amin(dim=2)  # min positive id (big = none)
    live = mx > 0
    # share[b, lc, c]: ranges overlap and both chunks hold a document
    share = (mx[:, None, :] >= mn[:, :, None]) & (mn[:, None, :] <= mx[:, :, None]) & live[:, None, :] & live[:, :, None]
    share = share.any(dim=0)                                            # [C (lc), C (c)]
    idx = torch.arange(C, device=doc_ids.device)
    earlier = idx[None, :] < idx[:, None]                               # c < lc
    per_chunk = (share & earlier) | torch.eye(C, dtype=torch.bool, device=doc_ids.device)
    r = torch.arange(cp, device=doc_ids.device)
    return per_chunk[r] | per_chunk[C - 1 - r]                          # rank r owns chunks r and 2cp-1-r


@dataclass
class ContextParallel:
    group: object
    T: int                     # global packed length
    emulate: tuple = None      # (cp, rank): ONE process plays rank `rank` of a cp-way group and moves nothing — the chunks a
                               # real run would receive are filled with copies of the rank's own first chunk (same bytes
                               # written, same kernels, same tile work: the document ids decide what the kernels do, not the
                               # values), the gradients a real run would send back are dropped.  `bench.py --emulate-rank`.

    def __post_init__(self):
        if self.emulate is not None:
            self.cp, self.rank = int(self.emulate[0]), int(self.emulate[1])
        else:
            self.cp = dist.get_world_size(self.group)
            self.rank = dist.get_rank(self.group)
        if self.T % (2 * self.cp * 128):
            raise ValueError(f"T={self.T} must be a multiple of 2*cp*128 = {2 * self.cp * 128}")
        self.Tc = self.T // (2 * self.cp)
        self._need = None                # bool [cp, 2cp] after set_documents(): rank r needs chunk c
        self._need_pending = None        # (pinned host tensor, event) while the device result is on its way
        self.halo_bytes = 0              # bytes this rank received through the halo exchange (diagnostics / tests)

    # ---- document-aware halo ---------------------------------------------------------------------------
    def set_documents(self, doc_ids: torch.Tensor) -> None:
        """doc_ids int [B, T] (0 = pad), identical on every rank: fixes which chunks each rank exchanges.
        Host tensor: exact `halo_need`.  Device tensor: `halo_need_ranges` on the device, the [cp, 2cp] table comes
        back through a pinned buffer without blocking (it is waited for at the first exchange of the step)."""
        if not doc_ids.is_cuda:
            # the RANGE criterion here too: the attention kernels decide per tile by id ranges, and the global K/V
            # buffers are uninitialised outside the received chunks — an exact (shared-id) table would be a strict
            # subset of what they may touch as soon as ids are not non-decreasing along a row
            self._need, self._need_pending = halo_need_ranges(doc_ids.detach(), self.cp).numpy().copy(), None
            return
        table = halo_need_ranges(doc_ids.detach(), self.cp)
        host = torch.empty(table.shape, dtype=torch.bool).pin_memory()
        host.copy_(table, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._need, self._need_pending = None, (host, ev)

    @property
    def need(self):
        if self._need_pending is not None:
            host, ev = self._need_pending
            if not ev.query():           # (a few hundred bytes issued when the batch arrived: long done by now; pass the
                ev.synchronize()         #  host copy of the ids to set_documents — bin/train.py does — and there is no event)
            self._need, self._need_pending = host.numpy().copy(), None
        return self._need

    @need.setter
    def need(self, value):
        self._need, self._need_pending = value, None

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

    def shard_audio(self, audio_positions, audio_output_lengths, rows_per_clip: int):
        """Qwen2-Audio under context parallelism (host side, numpy): which clips this rank's audio tower runs and where
        their rows go.  `audio_positions` int64 [sum(lengths)] are flat indices into the GLOBAL [B, T] token grid (clip
        after clip, `lengths[i]` consecutive entries each), `rows_per_clip` = rows the tower returns per clip (Ta).
        A clip belongs to every rank that owns at least one of its token positions (a clip straddling a chunk boundary
        runs on both sides: the tower is per clip, each side keeps its own rows and backpropagates only through them, so
        the parameter gradients still add up to the unsharded ones over the dp x cp reduce-scatter).
        Returns (clips, positions_local, rows): `clips` int64 [n_local] indices into the clip dim of `input_features`,
        `positions_local` flat indices into this rank's [B, 2*Tc] grid, `rows` int64 indices into the tower's
        [n_local * rows_per_clip] output rows, one per local position.  A rank without audio still gets clip 0 with
        zero rows kept, so that every rank runs the tower (FSDP2's collectives stay symmetric)."""
        import numpy as np
        pos = np.asarray(audio_positions, dtype=np.int64).reshape(-1)
        lens = np.asarray(audio_output_lengths, dtype=np.int64).reshape(-1)
        if int(lens.sum()) != pos.size:
            raise ValueError(f"audio positions ({pos.size}) and lengths (sum {int(lens.sum())}) disagree")
        b, col = pos // self.T, pos % self.T
        chunk = col // self.Tc
        lo, hi = self.my_chunks()
        mine = (chunk == lo) | (chunk == hi)
        local_col = np.where(chunk == lo, col - lo * self.Tc, col - hi * self.Tc + self.Tc)
        clip_of = np.repeat(np.arange(lens.size), lens)
        row_in_clip = np.arange(pos.size) - np.repeat(np.cumsum(lens) - lens, lens)
        clips = np.unique(clip_of[mine])
        if clips.size == 0:
            clips = np.zeros(1, dtype=np.int64)
        slot = np.full(lens.size, -1, dtype=np.int64)
        slot[clips] = np.arange(clips.size)
        positions_local = (b * 2 * self.Tc + local_col)[mine]
        rows = (slot[clip_of] * rows_per_clip + row_in_clip)[mine]
        return clips.astype(np.int64), positions_local.astype(np.int64), rows.astype(np.int64)

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


class _Transfer:
    """One batch of point-to-point transfers in flight + what to do with the staging buffers once it has landed."""

    def __init__(self, works, after):
        self.works, self.after = works, after

    def wait(self):
        for w in self.works:
            w.wait()                       # RCCL: the COMPUTE stream waits for the transfer; gloo: the host does
        for fn in self.after:
            fn()
        self.works, self.after = [], []


def _contig_chunk(t: torch.Tensor, start: int, n: int):
    v = t.narrow(1, start, n)
    return v if v.is_contiguous() else None


def _start_forward(cp: "ContextParallel", locals_):
    """locals_: list of [B, 2Tc, ...] tensors (K and V) -> (list of [B, T, ...] global tensors, _Transfer)."""
    Tc, me, need = cp.Tc, cp.rank, cp.need
    mine = cp.my_chunks()
    fulls = [x.new_empty((x.shape[0], cp.T) + tuple(x.shape[2:])) for x in locals_]
    ops, after = [], []
    for x, full in zip(locals_, fulls):
        for h, c in enumerate(mine):
            full.narrow(1, c * Tc, Tc).copy_(x.narrow(1, h * Tc, Tc))
        if cp.emulate is not None:                                # stand-ins for the chunks a real run would receive
            for c in range(2 * cp.cp):
                if need[me, c] and c not in mine:
                    full.narrow(1, c * Tc, Tc).copy_(x.narrow(1, 0, Tc))
                    cp.halo_bytes += x.narrow(1, 0, Tc).numel() * x.element_size()
            continue
        for p in range(cp.cp):
            if p == me:
                continue
            peer = dist.get_global_rank(cp.group, p)
            for h, c in enumerate(mine):                          # my chunks p needs
                if need[p, c]:
                    src = _contig_chunk(x, h * Tc, Tc)
                    ops.append(dist.P2POp(dist.isend, src if src is not None else x.narrow(1, h * Tc, Tc).contiguous(),
                                          peer, group=cp.group))
            for c in cp.my_chunks(p):                             # p's chunks I need
                if need[me, c]:
                    dst = _contig_chunk(full, c * Tc, Tc)
                    if dst is None:                               # B > 1: a chunk is strided in [B, T, ...]: stage it
                        stage = x.new_empty((x.shape[0], Tc) + tuple(x.shape[2:]))
                        after.append(lambda f=full, c=c, s=stage: f.narrow(1, c * Tc, Tc).copy_(s))
                        dst = stage
                    cp.halo_bytes += dst.numel() * dst.element_size()
                    ops.append(dist.P2POp(dist.irecv, dst, peer, group=cp.group))
    works = dist.batch_isend_irecv(ops) if ops else []
    return fulls, _Transfer(works, after)


def _start_backward(cp: "ContextParallel", g_fulls):
    """Partial dK/dV of the remote chunks go back to their owners; returns (list of local grads [B, 2Tc, ...] holding
    this rank's own contribution, _Transfer whose completion adds the peers' partial sums)."""
    Tc, me, need = cp.Tc, cp.rank, cp.need
    mine = cp.my_chunks()
    g_locals = [torch.cat([g.narrow(1, c * Tc, Tc) for c in mine], dim=1).contiguous() for g in g_fulls]
    ops, after = [], []
    for g, gl in zip(g_fulls, g_locals):
        if cp.emulate is not None:                                # (the partial sums of the stand-in chunks go nowhere)
            continue
        for p in range(cp.cp):
            if p == me:
                continue
            peer = dist.get_global_rank(cp.group, p)
            for c in cp.my_chunks(p):                             # chunks of p this rank used: their grads go back
                if need[me, c]:
                    src = _contig_chunk(g, c * Tc, Tc)
                    ops.append(dist.P2POp(dist.isend, src if src is not None else g.narrow(1, c * Tc, Tc).contiguous(),
                                          peer, group=cp.group))
            for h, c in enumerate(mine):                          # p used these chunks of mine
                if need[p, c]:
                    stage = gl.new_empty((gl.shape[0], Tc) + tuple(gl.shape[2:]))
                    after.append(lambda gl=gl, h=h, s=stage: gl.narrow(1, h * Tc, Tc).add_(s))
                    ops.append(dist.P2POp(dist.irecv, stage, peer, group=cp.group))
    works = dist.batch_isend_irecv(ops) if ops else []
    return g_locals, _Transfer(works, after)


class _Link:
    """Shared by the two autograd nodes of one layer's exchange."""

    def __init__(self, cp):
        self.cp, self.fwd, self.bwd, self.g_locals = cp, None, None, None


class _HaloStart(torch.autograd.Function):
    """forward: ISSUE the K/V halo exchange (returns the global buffers before the data has landed);
    backward: FINISH the dK/dV return exchange issued by _HaloFinish.backward."""

    @staticmethod
    def forward(ctx, k_local, v_local, link: _Link):
        fulls, link.fwd = _start_forward(link.cp, [k_local, v_local])
        ctx.link = link
        return fulls[0], fulls[1]

    @staticmethod
    def backward(ctx, _gk, _gv):
        link = ctx.link
        link.bwd.wait()
        gk, gv = link.g_locals
        link.bwd = link.g_locals = None
        return gk, gv, None


class _HaloFinish(torch.autograd.Function):
    """forward: WAIT for the K/V exchange right in front of the attention kernel;
    backward: ISSUE the return of the partial dK/dV (the query-path backward runs while they travel)."""

    @staticmethod
    def forward(ctx, k_full, v_full, link: _Link):
        link.fwd.wait()
        link.fwd = None
        ctx.link = link
        ctx.mark_dirty(k_full, v_full)
        return k_full, v_full

    @staticmethod
    def backward(ctx, gk_full, gv_full):
        link = ctx.link
        link.g_locals, link.bwd = _start_backward(link.cp, [gk_full.contiguous(), gv_full.contiguous()])
        return gk_full, gv_full, None


class _HaloReturn(torch.autograd.Function):
    """forward: nothing (the waiting is done INSIDE the split attention, between its local and its remote part);
    backward: ISSUE the return of the partial dK/dV (the query-path backward runs while they travel)."""

    @staticmethod
    def forward(ctx, k_full, v_full, link: _Link):
        ctx.link = link
        ctx.mark_dirty(k_full, v_full)
        return k_full, v_full

    @staticmethod
    def backward(ctx, gk_full, gv_full):
        link = ctx.link
        link.g_locals, link.bwd = _start_backward(link.cp, [gk_full.contiguous(), gv_full.contiguous()])
        return gk_full, gv_full, None


class KVExchange:
    """A layer's K/V halo exchange in flight.  `attend(q, mask, scale)` = the attention with the exchange hidden under
    its local part; calling the object (`finish()`) = the round-2 form: wait, then hand out the global (k_full, v_full)."""

    def __init__(self, cp, k_full, v_full, link):
        self.cp, self.k_full, self.v_full, self.link = cp, k_full, v_full, link

    def __call__(self):
        return _HaloFinish.apply(self.k_full, self.v_full, self.link)

    def attend(self, q_local, mask, scale=None):
        """Local query rows against (a) the keys of this rank's OWN chunks — already in the global buffers — while the
        remote chunks travel, then (b) the received chunks; (a) and (b) are merged by their log-sum-exp
        (functional.packed_attention_sharded_split; the reference's ring merges per-step (out, lse) pairs the same way:
        torch ... _context_parallel/_attention.py:182-183 via touchnet/utils/distributed.py:292-315)."""
        from touchnet_amd.models.backend import ops
        cp, link = self.cp, self.link
        k_full, v_full = _HaloReturn.apply(self.k_full, self.v_full, link)
        mine = cp.my_chunks()
        remote = [c for c in range(2 * cp.cp) if cp.need[cp.rank, c] and c not in mine]

        def wait():
            link.fwd.wait()
            link.fwd = None
        return ops().packed_attention_sharded_split(q_local, k_full, v_full, mask, cp.seq_shard(), cp.Tc, list(mine),
                                                    remote, wait, scale)


def exchange_kv(cp: ContextParallel, k_local: torch.Tensor, v_local: torch.Tensor) -> KVExchange:
    """Start the halo exchange of a layer's K/V.  Call the result's `attend(q, mask, scale)` AFTER the query-path work
    has been issued (or call it like a function to get the global (k_full, v_full) once they have landed).  Needs
    `cp.set_documents` (the all-gather fallback has nothing to overlap with: use `cp.gather_seq`)."""
    link = _Link(cp)
    k_full, v_full = _HaloStart.apply(k_local, v_local, link)
    return KVExchange(cp, k_full, v_full, link)


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
