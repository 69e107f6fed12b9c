"""Data parallelism for the packed path laid out for 288 GB of HBM: flat per-block buffers, ONE reduce-scatter and ONE
all-gather per block and step, optimizer state sharded (ZeRO-1 over buckets).

Role in the reference: `apply_fsdp` (touchnet/models/helper_func.py:134-202; FSDP2 `fully_shard` per block with
MixedPrecisionPolicy(param=bf16, reduce=fp32)) — which stays available behind `parallelize_fn` (models/parallelize.py)
and is what the tensor-parallel layout composes with.  Why a second engine: a 7-8 B model in bf16 is 17 GB of a 288 GB
device, so sharding the PARAMETERS buys nothing here, and FSDP2's per-parameter DTensor plumbing costs (measured on one
MI355X, profiles/r02_fsdp2_single_rank_tax.md: +108 ms on a 807 ms step) an fp32->bf16 cast into the all-gather input, a
copy-out of the gathered block into parameter tensors, a bf16->fp32 concatenating copy into the reduce-scatter input and
fp32 gradient shards — the latter two on every rank whatever the world size.  Here instead:

  * the bf16 parameters of a block are VIEWS into one flat buffer (`flat_p`); rank r owns the contiguous slice
    [r S, (r+1) S) of it: the in-place `all_gather_into_tensor(flat_p, flat_p[slice])` needs no copy-in and no copy-out;
  * a parameter's gradient is moved (one cast-copy, bf16 -> reduce dtype) into the block's flat staging buffer the moment
    autograd has accumulated it, and freed; when the block's last gradient has arrived `reduce_scatter_tensor` runs on a
    side stream under the rest of the backward; staging buffers are a small pool (a block's buffer is reused two blocks
    later), the reduced shards are persistent;
  * the optimizer (utils/optimizer.FusedAdamW: fp32 master + moments) sees ONE tensor per block — the rank's slice of
    `flat_p` with the reduced gradient shard — and rewrites the bf16 slice in the same pass; the global norm is the
    all-reduced sum over the shards;
  * the all-gathers of the updated slices are issued right after the optimizer step on the side stream; a block's forward
    waits (stream-side, no host sync) for its own buffer only, so they travel under the next step's frontend and forward.

Memory per GPU at N ranks for P parameters: 2 P (bf16) + 12 P / N (fp32 master, m, v) + the staging pool, against
16 P unsharded: Qwen2-Audio-7B at N = 8: 17 + 12.6 + 8 GB.  Gradient averaging, the fp32 reduction and the
skip-on-nonfinite semantics are FSDP2's / the reference's; `reduce_dtype=bfloat16` halves the xGMI bytes.
"""
from __future__ import annotations

import os
import weakref
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from touchnet_amd.models.helper_func import block_groups

ALIGN = 128          # elements: every parameter starts on a 256-byte boundary inside its flat buffer


class Shard:
    """What the optimizer sees of a bucket: this rank's slice of the flat parameter buffer + its reduced gradient.  (Not
    an nn.Parameter: the gradient shard is in the REDUCE dtype, which torch does not allow to differ from the data's.)
    `device_mesh` / `placements` / `shape` / `stride()` describe it as a dim-0 sharded 1-D DTensor for DCP
    (FusedAdamW._as_saved)."""
    requires_grad = True

    def __init__(self, data: torch.Tensor, total: int, mesh=None):
        self.data, self.grad = data, None
        self.shape = torch.Size((total,))
        self.device_mesh = mesh
        if mesh is not None:
            from torch.distributed.tensor import Shard as _S
            self.placements = (_S(0),)

    def stride(self):
        return (1,)

    def numel(self):
        return self.data.numel()

    device = property(lambda self: self.data.device)
    dtype = property(lambda self: self.data.dtype)


class EmulatedShardMesh:
    """Stand-in mesh for `FlatShardedDataParallel`: one process as rank `rank` of a `size`-rank data-parallel group."""
    emulated = True

    def __init__(self, size: int, rank: int = 0):
        self._size, self.rank = int(size), int(rank)

    def size(self, dim=None):
        return self._size

    def get_group(self, dim=None):
        return None


class _Bucket:
    def __init__(self, name: str, params: List[nn.Parameter], world: int, rank: int, mesh):
        self.name, self.params = name, params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        # The padded length does not depend on the world size for every world that divides 64 (else: their least common
        # multiple): the optimizer state of a bucket is saved as a 1-D DTensor of this length sharded on dim 0, and a
        # checkpoint written by N ranks can then be resharded by DCP onto M (it was padded to world * ALIGN)
        import math
        unit = ALIGN * (64 * world // math.gcd(64, world))
        self.total = (off + unit - 1) // unit * unit
        self.S = self.total // world
        # gaps between parameters + the tail: staging bytes no gradient ever lands on (kept zero)
        self.gaps = [(o + p.numel(), (self.offsets[i + 1] if i + 1 < len(params) else self.total))
                     for i, (o, p) in enumerate(zip(self.offsets, params))]
        self.gaps = [(a, b) for a, b in self.gaps if b > a]
        p0 = params[0]
        self.flat_p = torch.zeros(self.total, dtype=p0.dtype, device=p0.device)
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.shard = Shard(self.flat_p[rank * self.S:(rank + 1) * self.S], self.total, mesh)
        self.gshard: Optional[torch.Tensor] = None      # reduced gradient shard (persistent, allocated on first use)
        self.stage: Optional[torch.Tensor] = None       # staging buffer of this step (from the pool)
        self.arrived = [False] * len(params)
        self.pending = len(params)
        self.launched = False
        self.opened = False                              # a staging buffer has been acquired for this step
        self.reduced = None                             # event: the reduce-scatter has finished
        self.params_ready = None                        # event: the all-gather of the updated slices has finished


class FlatShardedDataParallel:
    def __init__(self, model: nn.Module, mesh, reduce_dtype: torch.dtype = torch.float32, replicate_mesh=None):
        """`mesh`: the 1-D device mesh gradients are averaged over (dp, or dp x cp flattened).  `replicate_mesh` (HSDP,
        the reference's `dp_replicate` dimension, touchnet/utils/distributed.py:139-157): a second 1-D mesh over which
        whole replicas of the sharded state exist — the reduced gradient shard of a bucket is all-reduced (AVG) over it
        right behind the reduce-scatter, so every replica's optimizer sees the same shard.  Buckets: one per
        transformer block (models.helper_func.block_groups, the reference's FSDP units) + one per module that owns any of
        the remaining parameters.  Inside a bucket that does take part, a parameter without a gradient counts as a zero
        gradient (torch would skip it; no such parameter exists in the models of this path)."""
        self.model, self.mesh = model, mesh
        # `EmulatedShardMesh(N, r)`: ONE process plays rank r of an N-rank group (bench.py --emulate-shards): buckets,
        # shard sizes, optimizer state and the staging pool are a real rank's; the reduce-scatter is replaced by a copy of
        # this rank's slice, the all-gather is skipped (the other slices go stale: timing / memory emulation only)
        self.emulated = bool(getattr(mesh, "emulated", False))
        if self.emulated:
            self.group, self.world, self.rank = None, mesh.size(), mesh.rank
        else:
            self.group = mesh.get_group()
            self.world, self.rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        self.rep_group = None
        if replicate_mesh is not None and replicate_mesh.size() > 1:
            self.rep_group = replicate_mesh.get_group()
        self.reduce_dtype = reduce_dtype
        # One rank (TN_FORCE_FSDP=1 on a single GPU): reduce-scatter and all-gather are the identity, so the hooks write
        # straight into the persistent gradient shard and no collective is issued — RCCL would run its generic one-rank
        # kernel instead (oneRankReduce<PreMulSum>, measured 48.5 ms per step for 34 GB: 1.4 TB/s), which says nothing
        # about what a rank pays at N > 1.  TN_DP_FORCE_COLLECTIVES=1 issues them anyway (the 1-rank RCCL test).
        self.identity = (self.world == 1 and self.rep_group is None
                         and os.environ.get("TN_DP_FORCE_COLLECTIVES") != "1")
        params = [p for p in model.parameters() if p.requires_grad]
        self.device = params[0].device
        self.cuda = self.device.type == "cuda"
        seen, self.buckets, self._hooked_modules = set(), [], []
        blocks = [(f"{gi}.{bi}", blk) for gi, grp in enumerate(block_groups(model)) for bi, blk in enumerate(grp)]
        for name, blk in blocks:
            ps = [p for p in blk.parameters() if p.requires_grad and id(p) not in seen]
            seen.update(id(p) for p in ps)
            if ps:
                self.buckets.append(_Bucket(f"block.{name}", ps, self.world, self.rank, None if self.emulated else mesh))
                self._hooked_modules.append((blk, self.buckets[-1]))
        # the remaining parameters: one bucket per owning module (embedding, final norm, each head, projector, stem ...).
        # A bucket is all-or-nothing for the optimizer: one NONE of whose parameters took part in a step is skipped like
        # torch skips `grad is None` parameters (no weight decay either) — Kimi-Audio's audio head and mimo norm in a
        # text-head step — so parameters of different modules must not share one.
        self.rest = []
        for mname, mod in model.named_modules():
            ps = [p for p in mod.parameters(recurse=False) if p.requires_grad and id(p) not in seen]
            seen.update(id(p) for p in ps)
            if ps:
                self.rest.append(_Bucket(f"rest.{mname or 'root'}", ps, self.world, self.rank, None if self.emulated else mesh))
        self.buckets += self.rest
        self._of = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._of[id(p)] = (b, i)
                p.register_post_accumulate_grad_hook(self._on_grad)
        # every replica starts from rank 0's numbers (the seeds agree already; this makes it unconditional)
        for b in self.buckets:
            if not self.emulated:
                if self.rep_group is not None:           # (replica 0's shard-rank-0 numbers reach everybody)
                    dist.broadcast(b.flat_p, src=dist.get_global_rank(self.rep_group, 0), group=self.rep_group)
                dist.broadcast(b.flat_p, src=dist.get_global_rank(self.group, 0), group=self.group)
        # HIGH priority: ROCm multiplexes HIP streams onto a few hardware queues, and two streams of one priority may share
        # a queue — their kernels then run strictly one after the other (measured, round 5: every reduce-scatter of a step
        # on the queue of the backward's GEMMs, 0 % of the RCCL kernels' time beside compute).  Streams of another priority
        # get a queue of their own (profiles/r05o_*).  TN_COMM_STREAM_PRIORITY=0 restores the default stream.
        prio = int(os.environ.get("TN_COMM_STREAM_PRIORITY", "-1"))
        self.comm = torch.cuda.Stream(device=self.device, priority=prio) if self.cuda else None
        self._pool, self._pool_free_at = [], []          # staging buffers + the event after which each is reusable
        self._pool_numel = max(b.total for b in self.buckets if b not in self.rest) if len(self.rest) < len(self.buckets) else 0
        for blk, b in self._hooked_modules:
            blk.register_forward_pre_hook(lambda m, a, _b=b: self._wait_params(_b))

        def wait_rest(module, args):
            for b in self.rest:
                self._wait_params(b)
        model.register_forward_pre_hook(wait_rest)
        if hasattr(model, "register_state_dict_pre_hook"):
            model.register_state_dict_pre_hook(lambda module, prefix, keep_vars: self.wait_params())
        self.avg = dist.ReduceOp.AVG
        self.staged_bytes = 0                            # diagnostics (tests): bytes moved into staging this step
        # the blocks' linear weights (used once per step, by our own GEMM nodes): their weight-gradient GEMMs write the
        # staging views themselves
        self.sunk = 0
        if self.cuda:
            from touchnet_amd import functional as F
            F.WGRAD_RETURNS_NEED_SYNC = True             # (see functional._beside)
        if self.cuda and os.environ.get("TN_DP_GRAD_SINKS", "1") != "0":
            from touchnet_amd import functional as F
            for blk, b in self._hooked_modules:
                for m in blk.modules():
                    w = getattr(m, "weight", None)
                    if isinstance(m, nn.Linear) and w is not None and id(w) in self._of and w.dim() == 2:
                        F.GRAD_SINKS[id(w)] = weakref.ref(self)
                        self.sunk += 1

    # ------------------------------------------------------------------ optimizer view
    def named_shards(self):
        return [(b.name, b.shard) for b in self.buckets]

    # ------------------------------------------------------------------ gradients
    def _acquire(self, b: _Bucket) -> torch.Tensor:
        if self.identity:
            if b.gshard is None:
                b.gshard = torch.zeros(b.S, dtype=self.reduce_dtype, device=self.device)
            return b.gshard
        if b in self.rest:                               # open during the whole backward: a buffer of its own
            if b.stage is None:
                b.stage = torch.zeros(b.total, dtype=self.reduce_dtype, device=self.device)
            if b.reduced is not None and self.cuda:
                self._wait_writers(b.reduced)
            return b.stage
        for i, ev in enumerate(self._pool_free_at):
            if ev is not False:                          # False = in use by an open bucket
                if ev is not None and self.cuda:
                    self._wait_writers(ev)
                self._pool_free_at[i] = False
                b._slot = i
                return self._pool[i][:b.total]
        self._pool.append(torch.zeros(self._pool_numel, dtype=self.reduce_dtype, device=self.device))
        self._pool_free_at.append(False)
        b._slot = len(self._pool) - 1
        return self._pool[-1][:b.total]

    def _wait_writers(self, ev) -> None:
        """Every stream that may write gradients into a staging buffer waits for its previous reduce-scatter: the caller's,
        and — with weight-gradient GEMMs on a side stream — that stream AND the backward's own (a bucket may be opened
        by a GEMM on the side stream while norm weights / biases are cast-copied on the main one)."""
        from touchnet_amd import functional as F
        for st in F.wgrad_streams():
            st.wait_event(ev)

    def take(self, p):
        """-> (this parameter's [shape] view in its bucket's staging buffer, already holds a partial gradient?).  Called by
        the post-accumulate hook below and — for the linear layers' weights of the blocks (`functional.GRAD_SINKS`) — by
        the weight-gradient GEMM itself, which then writes the view directly (no cast-copy)."""
        b, i = self._of[id(p)]
        if b.launched:
            raise RuntimeError(f"gradient of a parameter of bucket {b.name} arrived after its reduce-scatter was issued "
                               "(gradient accumulation over several backward passes is not supported by this engine)")
        if not b.opened:                                # (several views may be taken before the first `done`: a grouped
            b.opened = True                              #  weight-gradient launch writes three of them at once)
            b.stage = self._acquire(b)
            for a, e in b.gaps:                          # (stale numbers of the buffer's previous user)
                b.stage[a:e].zero_()
        o = b.offsets[i]
        return b.stage[o:o + p.numel()].view(p.shape), b.arrived[i]

    def owns(self, p) -> bool:
        e = self._of.get(id(p))
        return e is not None and e[0].params[e[1]] is p      # (ids can be recycled by later objects)

    def done(self, p) -> None:
        b, i = self._of[id(p)]
        if not b.arrived[i]:
            b.arrived[i] = True
            b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _on_grad(self, p: nn.Parameter) -> None:
        if p.grad is None:        # (autograd also runs the hook for a weight whose GEMM wrote the staging view itself and
            return                #  returned no gradient)
        view, partial = self.take(p)
        if partial:
            view.add_(p.grad)
        else:
            view.copy_(p.grad)                           # the one cast-copy of this gradient
        self.staged_bytes += p.numel() * (p.grad.element_size() + view.element_size())
        p.grad = None
        self.done(p)

    def _launch(self, b: _Bucket) -> None:
        if self.identity:
            b.launched = True
            return
        if b.gshard is None:
            b.gshard = torch.empty(b.S, dtype=self.reduce_dtype, device=self.device)
        if self.cuda:
            from touchnet_amd import functional as F
            for st in F.wgrad_streams():                 # everything that wrote this bucket's views, wherever it ran
                self.comm.wait_stream(st)
            with torch.cuda.stream(self.comm):
                if self.emulated:
                    b.gshard.copy_(b.stage[self.rank * b.S:(self.rank + 1) * b.S])
                else:
                    dist.reduce_scatter_tensor(b.gshard, b.stage, op=self.avg, group=self.group)
                    if self.rep_group is not None:
                        dist.all_reduce(b.gshard, op=self.avg, group=self.rep_group)
                b.reduced = torch.cuda.Event()
                b.reduced.record()
            if b not in self.rest:
                self._pool_free_at[b._slot] = b.reduced
        else:
            if self.emulated:
                b.gshard.copy_(b.stage[self.rank * b.S:(self.rank + 1) * b.S])
            else:
                dist.reduce_scatter_tensor(b.gshard, b.stage, op=self.avg, group=self.group)
                if self.rep_group is not None:
                    dist.all_reduce(b.gshard, op=self.avg, group=self.rep_group)
            if b not in self.rest:
                self._pool_free_at[b._slot] = None
        b.launched = True

    def finish_backward(self) -> None:
        """After `loss.backward()`: issue what is still open (blocks some of whose parameters took no part in this step
        get zeros there; blocks NONE of whose parameters did are skipped on every rank alike and keep `grad = None`),
        make the compute stream wait for the reductions and hand the shards to the optimizer."""
        for b in self.buckets:
            if not b.launched and any(b.arrived):
                for i, (p, o) in enumerate(zip(b.params, b.offsets)):
                    if not b.arrived[i]:
                        b.stage[o:o + p.numel()].zero_()
                self._launch(b)
        for b in self.buckets:
            if b.launched:
                if self.cuda and b.reduced is not None:
                    torch.cuda.current_stream().wait_event(b.reduced)
                b.shard.grad = b.gshard
            else:
                b.shard.grad = None

    def zero_grad(self) -> None:
        for b in self.buckets:
            if b.opened and not b.launched and not self.identity and b not in self.rest:
                self._pool_free_at[b._slot] = None        # (a step that died inside its backward: give the buffer back)
            b.arrived = [False] * len(b.params)
            b.pending, b.launched, b.opened = len(b.params), False, False
            b.shard.grad = None
            for p in b.params:
                p.grad = None
        self.staged_bytes = 0

    # ------------------------------------------------------------------ parameters
    def gather_params(self) -> None:
        """After the optimizer step (it rewrote this rank's slice of every flat buffer): all-gather the slices in the order
        the next forward needs them, on the side stream."""
        if self.identity or self.emulated:
            return
        order = self.rest + [b for b in self.buckets if b not in self.rest]
        if self.cuda:
            done = torch.cuda.Event()
            done.record()
            self.comm.wait_event(done)
            with torch.cuda.stream(self.comm):
                for b in order:
                    dist.all_gather_into_tensor(b.flat_p, b.shard.data, group=self.group)
                    b.params_ready = torch.cuda.Event()
                    b.params_ready.record()
        else:
            for b in order:
                dist.all_gather_into_tensor(b.flat_p, b.shard.data, group=self.group)

    def wait_params(self) -> None:
        """The current stream waits for every pending all-gather of updated slices.  The blocks' and the model's forward
        pre-hooks do that per bucket; anything ELSE that reads parameters right behind a step (state_dict / checkpoint
        save, evaluation through a submodule, an EMA) must call this first — a state_dict pre-hook does it for
        `model.state_dict()`."""
        for b in self.buckets:
            self._wait_params(b)

    def _wait_params(self, b: _Bucket) -> None:
        if self.cuda and b.params_ready is not None:
            torch.cuda.current_stream().wait_event(b.params_ready)
            b.params_ready = None


# ---------------------------------------------------------------------------------------------------------------------
# The engine behind the reference's own hooks (touchnet/bin/train.py).  Its trainer calls, in this order
# (tests/golden/boundary.json `model_setup_sequence` / `train_step_sequence`):
#     model = parallelize_fn(model ON THE META DEVICE, world_mesh, parallel_dims, job)       :259
#     model.to_empty(device); model.post_init(); additional_post_init_fn(...); model.to(float32)   :274-283
#     optimizers = build_optimizers_fn([model], job);  lr_schedulers = build_lr_schedulers_fn(optimizers, job)   :296-297
#     per step:  optimizers.zero_grad();  forward;  loss.backward();  clip_grad_norm_(model parameters, max_norm);
#                optimizers.step()  (or optimizers.zero_grad() on a non-finite norm);  lr_schedulers.step()      :396-474
# The flat buffers cannot exist while the model is on the meta device, so `parallelize_fn` only MARKS the model
# (`mark_flat_engine`) and `build_optimizers_fn` — the first hook that sees real tensors — builds the engine
# (`build_flat_engine_optimizer`): parameters go to the compute dtype (`training_mixed_precision_param`, what FSDP2's
# MixedPrecisionPolicy would all-gather in), the float32 values the trainer just created become the optimizer's master
# copy.  After `backward()` the parameters carry no `.grad` (every gradient sits in a reduce-scatter input), so the
# trainer's own clip_grad_norm_ computes 0 and scales nothing; norm, clip and the skip on a non-finite norm happen on
# the device inside `step()` (utils/optimizer.FusedAdamW), whose result is kept in `last_grad_norm`.
# ---------------------------------------------------------------------------------------------------------------------
def mark_flat_engine(model: nn.Module, mesh, reduce_dtype: torch.dtype, param_dtype: torch.dtype,
                     replicate_mesh=None) -> None:
    model._tn_flat_dp = {"mesh": mesh, "reduce_dtype": reduce_dtype, "param_dtype": param_dtype,
                         "replicate_mesh": replicate_mesh}


class FlatEngineOptimizer:
    """The `optimizers` object of the reference trainer for a model on the flat engine (see above)."""

    def __init__(self, engine: FlatShardedDataParallel, inner):
        self.engine, self.inner = engine, inner
        self.last_grad_norm = None

    lr = property(lambda self: self.inner.lr, lambda self, v: setattr(self.inner, "lr", v))

    def zero_grad(self, *args, **kwargs) -> None:
        self.inner.zero_grad()
        self.engine.zero_grad()

    def step(self, lr=None):
        from touchnet_amd.models.backend import ops as _ops
        if hasattr(_ops(), "sync_wgrad_stream"):
            _ops().sync_wgrad_stream()
        self.engine.finish_backward()          # reduce-scatters ran under the backward; shards go to the optimizer
        self.last_grad_norm = self.inner.step(lr) if lr is not None else self.inner.step()
        self.engine.gather_params()            # all-gathers of the updated slices travel under the next forward
        return self.last_grad_norm

    def state_dict(self):
        return self.inner.state_dict()

    def load_state_dict(self, sd) -> None:
        self.inner.load_state_dict(sd)
        # the bf16 slices were rewritten from the loaded masters on this rank only: every rank needs every slice
        self.engine.gather_params()
        self.engine.wait_params()



def _check_replicas_drew_the_same_values(tensors, group) -> None:
    """Start-up check of the flat engine: every rank of `group` must hold the SAME float32 initial values (the bf16 buffers
    are broadcast from rank 0, the optimizer's master copies come from each rank's own draw).  Per tensor three float64
    statistics are compared — sum, sum of squares and a POSITION-WEIGHTED sum (weights 1 .. 8191 repeating) — with one MIN /
    MAX all-reduce of the stacked vector: differences that cancel across tensors, inside a tensor, or a permutation of a
    tensor's elements no longer pass as they did with a single grand total; non-finite values are reported as
    such instead of as a rank mismatch."""
    if not tensors:
        return
    def three(t):
        d = t.reshape(-1).double()
        w = (torch.arange(d.numel(), device=d.device) % 8191 + 1).double()
        return torch.stack([d.sum(), (d * d).sum(), (d * w).sum()])
    stats = torch.stack([three(t) for t in tensors]).reshape(-1)
    if not bool(torch.isfinite(stats).all()):
        raise RuntimeError("flat data-parallel engine: non-finite values in the initial parameters of this rank "
                           "(the replica check cannot compare them): fix the initialisation")
    hi_, lo_ = stats.clone(), stats.clone()
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN, group=group)
    if not torch.equal(hi_, lo_):
        bad = int((hi_ != lo_).nonzero()[0]) // 3
        raise RuntimeError("flat data-parallel engine: the ranks hold different initial parameter values (tensor "
                           f"{bad} of {len(tensors)}: its checksums differ across ranks): seed every rank alike "
                           "or broadcast the model before build_optimizers_fn")

def build_flat_engine_optimizer(model: nn.Module, make_optimizer):
    """`make_optimizer(named_shards, process_group)` -> the optimizer over the engine's shards.  Returns the
    FlatEngineOptimizer; the engine is also left at `model._tn_flat_engine`."""
    mark = model._tn_flat_dp
    pdt = mark["param_dtype"]
    # float32 values of the parameters that are about to be rounded to the compute dtype (same storage as the
    # parameters hold now: no copy until the cast below allocates the new tensors)
    keep = {}
    with torch.no_grad():
        for p in model.parameters():
            if p.requires_grad and p.is_floating_point() and p.dtype != pdt:
                keep[id(p)] = p.data
                p.data = p.data.to(pdt)
    engine = FlatShardedDataParallel(model, mark["mesh"], reduce_dtype=mark["reduce_dtype"],
                                     replicate_mesh=mark.get("replicate_mesh"))
    inner = make_optimizer(engine.named_shards(), mark["mesh"].get_group())
    masters = getattr(inner, "state", None)
    if keep and isinstance(masters, list) and len(masters) == len(engine.buckets):
        # the optimizer drew its master copies from the rounded slices: give them the trainer's float32 numbers
        with torch.no_grad():
            for b, st in zip(engine.buckets, masters):
                m = st.get("master") if isinstance(st, dict) else None
                if m is None or m.dtype != torch.float32 or m.data_ptr() == b.shard.data.data_ptr():
                    continue
                lo, hi = engine.rank * b.S, (engine.rank + 1) * b.S
                for p, o in zip(b.params, b.offsets):
                    src = keep.get(id(p))
                    a, e = max(lo, o), min(hi, o + p.numel())
                    if src is not None and e > a:
                        m[a - lo:e - lo].copy_(src.reshape(-1)[a - o:e - o])
        # The bf16 buffers were broadcast from rank 0 by the engine; the masters above come from THIS rank's float32 draw.
        # They agree only if every rank drew the same values (the Trainer seeds all ranks alike).  A rank whose draw
        # differs would train a replica that drifts away silently: compare a checksum of the float32 values over the
        # group (one tiny all-reduce at start-up) and refuse loudly.
        group = mark["mesh"].get_group()
        if engine.world > 1 and not engine.emulated:
            _check_replicas_drew_the_same_values(list(keep.values()), group)
    model._tn_flat_engine = engine
    return FlatEngineOptimizer(engine, inner)
