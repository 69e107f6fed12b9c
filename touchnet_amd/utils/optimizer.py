"""Fused clip + AdamW on the HIP kernels (touchnet_amd/csrc/optim.hip).

Hyper-parameters and update rule are the reference's (touchnet/utils/optimizer.py:157-172:
AdamW, betas (0.9, 0.95), weight decay 0.1 on every parameter, eps 1e-8) plus the global-norm clip of
touchnet/utils/distributed.py:426-491 and the skip-on-nonfinite of touchnet/bin/train.py:467-473 — all
evaluated on the device, so the optimizer step issues no host synchronisation: THREE launches per gradient dtype
(multi-tensor sum of squares, its final reduction, multi-tensor AdamW) plus one 1-thread launch that turns the
norm into the step state (step count — advanced only on a finite norm, like torch's AdamW under the reference's
skip —, bias corrections, clip coefficient).

Precision layout (identical to FSDP2's MixedPrecisionPolicy(param=bf16, reduce=fp32) that the reference
applies, touchnet/models/helper_func.py:165):
  * single GPU:  module parameters are bf16 (what the kernels read); this class owns the fp32 master
                 copy and the fp32 Adam moments, and rewrites the bf16 parameter in the same pass
  * FSDP2:       the sharded parameters ARE the fp32 masters (FSDP all-gathers bf16 copies); the kernel
                 runs on each rank's local shard and the squared norm is all-reduced over the mesh dimensions the
                 parameters are SHARDED on (found from the DTensor placements when no group is passed)
"""
from __future__ import annotations

import os
from typing import Any, Dict, Iterable, List, Optional

import torch

from touchnet_amd import _C


def _local(t: torch.Tensor) -> torch.Tensor:
    return t._local_tensor if hasattr(t, "_local_tensor") else t


def _shard_groups(params) -> List[Any]:
    """Process groups the squared gradient norm has to be summed over: one per mesh dimension on which the DTensor
    parameters are sharded (replicated dimensions hold identical gradients after FSDP/HSDP's reduction)."""
    groups, seen = [], set()
    for p in params:
        mesh = getattr(p, "device_mesh", None)
        if mesh is None:
            continue
        for d, pl in enumerate(p.placements):
            if pl.is_shard() and mesh.size(d) > 1 and (id(mesh), d) not in seen:
                seen.add((id(mesh), d))
                groups.append(mesh.get_group(d))
    return groups


class FusedAdamW:
    def __init__(self, params: Iterable, lr=8e-4, betas=(0.9, 0.95), eps=1e-8,
                 weight_decay=0.1, max_norm: float = 1.0, process_group=None, tp_group=None, tp_param_ids=()):
        """`params`: parameters, or `(name, parameter)` pairs (`model.named_parameters()`): the names key the checkpoint
        state (`state_dict()`), so pass them whenever the optimizer is checkpointed.
        `tp_group` / `tp_param_ids` (models.tensor_parallel.tp_param_ids): parameters sharded over the tensor-parallel
        group — their squared gradient norm is summed over it, the replicated parameters' is not."""
        items = list(params)
        if items and isinstance(items[0], (tuple, list)):
            items = [(n, p) for n, p in items if p.requires_grad]
            self.names = [n.replace("_checkpoint_wrapped_module.", "") for n, _ in items]
            self.params = [p for _, p in items]
        else:
            self.params = [p for p in items if p.requires_grad]
            self.names = [f"param.{i}" for i in range(len(self.params))]
        if len(set(self.names)) != len(self.names):
            raise ValueError("FusedAdamW: parameter names must be unique")
        self.tp_group = tp_group
        self._tp_index = {i for i, p in enumerate(self.params) if id(p) in set(tp_param_ids)}
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        # explicit group (bin/train.py: the flattened dp x cp mesh) or the groups read off the DTensor placements
        self.groups = [process_group] if process_group is not None else _shard_groups(self.params)
        dev = _local(self.params[0]).device
        self.state = []
        for p in self.params:
            lp = _local(p.data)
            master = lp if lp.dtype == torch.float32 else lp.detach().float().clone()
            self.state.append(dict(master=master, m=torch.zeros_like(master), v=torch.zeros_like(master)))
        self.norm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_state = torch.zeros(8, dtype=torch.float32, device=dev)   # tn_adamw_prepare's device state
        self._partial, self._keep = None, None
        self._pipe = None                     # (groups, group of parameter i, side stream, events): pipeline_updates()
        self._pipe_delay_cycles = 0           # tests: hold the update stream back so that a missing wait shows
        # workgroups of a side-stream update launch (tn_adamw_multi_bounded): few enough to leave the forward's kernels the
        # machine, enough to finish under it
        self.side_workgroups = int(os.environ.get("TN_ADAMW_SIDE_WORKGROUPS", "128"))

    # ------------------------------------------------------------------ updates under the next forward
    def pipeline_updates(self, groups: List[List[int]]) -> None:
        """Run the AdamW launches on a SIDE stream, one launch per group of parameter indices (in the order given), and
        record one event per group: the caller makes the consumer of group g wait for it (`wait_group`).  The update is
        HBM-bound (28 B per parameter), the forward that follows is MFMA-bound: issued like this the update of block
        i+1… runs under the forward of block i instead of in front of it.  The norm / clip state is computed on the
        calling stream first, so the arithmetic is unchanged.  Plain (unsharded) parameters only."""
        if any(hasattr(p, "_local_tensor") for p in self.params):
            raise ValueError("pipeline_updates: sharded parameters are updated by their data-parallel engine's schedule")
        seen = sorted(i for g in groups for i in g)
        if seen != list(range(len(self.params))):
            raise ValueError("pipeline_updates: the groups must cover every parameter exactly once")
        of = {i: gi for gi, g in enumerate(groups) for i in g}
        self._pipe = ([list(g) for g in groups], of, torch.cuda.Stream(),
                      [torch.cuda.Event() for _ in groups], list(range(len(groups))))

    def set_launch_order(self, order: List[int]) -> None:
        """order of the groups' launches on the side stream (default: as given) — the order the forward consumes them in"""
        if sorted(order) != list(range(len(self._pipe[0]))):
            raise ValueError("set_launch_order: a permutation of the group indices")
        self._pipe[4][:] = list(order)

    def wait_group(self, gi: int) -> None:
        """the calling stream waits for the last update of group `gi` (no-op before the first step)"""
        if self._pipe is not None:
            torch.cuda.current_stream().wait_event(self._pipe[3][gi])

    def wait_updates(self) -> None:
        """the calling stream waits for every pending update (checkpoints, evaluation, anything outside the hooks)"""
        if self._pipe is not None:
            torch.cuda.current_stream().wait_stream(self._pipe[2])

    # ------------------------------------------------------------------ multi-tensor tables
    @staticmethod
    def _table(rows, device):
        t = torch.tensor(rows, dtype=torch.int64).pin_memory()
        return t.to(device, non_blocking=True)

    def _sumsq(self, by_dtype):
        """norm_sq += sum(g^2) over all gradients: one multi-tensor launch per gradient dtype (the gradient
        tensors are new allocations every step, so the pointer table is rebuilt and uploaded each time: ~20 KB)."""
        lib, p_, st = _C.lib(), _C.ptr, _C.stream
        chunk = int(lib.tn_sumsq_multi_chunk())
        keep = []
        for dt, items in by_dtype.items():
            gs = [g for _, g in items]
            sizes = [g.numel() for g in gs]
            first, tot = [], 0
            for n in sizes:
                first.append(tot)
                tot += (n + chunk - 1) // chunk
            table = self._table([[g.data_ptr() for g in gs], sizes, first], gs[0].device)
            if self._partial is None or self._partial.numel() < tot:
                self._partial = torch.empty(tot, dtype=torch.float32, device=gs[0].device)
            _C.check(lib.tn_sumsq_multi(p_(table[0]), p_(table[1]), p_(table[2]), len(gs), tot, p_(self._partial),
                                        p_(self.norm_sq), _C.dcode(gs[0]), st()), "tn_sumsq_multi")
            keep.append((table, gs))
        return keep

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self, lr: Optional[float] = None) -> torch.Tensor:
        """One clip + AdamW step.  Returns the (pre-clip) global grad norm as a 0-d device tensor."""
        lib, p_, st = _C.lib(), _C.ptr, _C.stream
        lr = self.lr if lr is None else lr
        b1, b2 = self.betas
        self.wait_updates()                   # (pipelined: the last step's launches read what `_keep` is about to release)
        self.norm_sq.zero_()
        by_dtype: Dict[torch.dtype, list] = {}
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            g = _local(p.grad).contiguous()
            if g.numel():
                by_dtype.setdefault(g.dtype, []).append((i, g))
        if self.tp_group is not None and self._tp_index:
            # sum over the tp ranks for the tp-sharded parameters only, then add the replicated ones' (counted once)
            pick = lambda inside: {dt: [it for it in items if (it[0] in self._tp_index) == inside]
                                   for dt, items in by_dtype.items()}
            drop_empty = lambda d: {dt: v for dt, v in d.items() if v}
            keep = self._sumsq(drop_empty(pick(True)))
            from touchnet_amd.models.tensor_parallel import tp_all_reduce
            tp_all_reduce(self.norm_sq, self.tp_group)
            keep += self._sumsq(drop_empty(pick(False)))
        else:
            keep = self._sumsq(by_dtype)
        for grp in self.groups:
            torch.distributed.all_reduce(self.norm_sq, group=grp)
        _C.check(lib.tn_adamw_prepare(p_(self.norm_sq), p_(self.step_state), b1, b2, float(self.max_norm), st()),
                 "tn_adamw_prepare")
        chunk = int(lib.tn_adamw_multi_chunk())
        # launches: one per gradient dtype — or, pipelined, per (group, dtype) on the side stream; ONE table upload either way
        if self._pipe is None:
            launches = [(None, items) for items in by_dtype.values()]
        else:
            groups, of = self._pipe[0], self._pipe[1]
            launches = []
            for gi in self._pipe[4]:
                for items in by_dtype.values():
                    sub = [it for it in items if of[it[0]] == gi]
                    launches.append((gi, sub))
        rows = [[], [], [], [], [], [], []]
        spans = []
        for gi, items in launches:
            tot, a = 0, len(rows[0])
            for i, g in items:
                s = self.state[i]
                lp = _local(self.params[i].data)
                rows[0].append(s["master"].data_ptr())
                rows[1].append(s["m"].data_ptr())
                rows[2].append(s["v"].data_ptr())
                rows[3].append(g.data_ptr())
                rows[4].append(lp.data_ptr() if lp.dtype == torch.bfloat16 else 0)
                rows[5].append(g.numel())
                rows[6].append(tot)
                tot += (g.numel() + chunk - 1) // chunk
            spans.append((gi, a, len(rows[0]), tot, _C.dcode(items[0][1]) if items else 0))
        if rows[0]:
            t = self._table(rows, _local(self.params[0]).device)
            keep.append(t)

            bound = 0 if self._pipe is None else self.side_workgroups

            def launch(a, b, tot, code):
                _C.check(lib.tn_adamw_multi_bounded(p_(t[0][a:b]), p_(t[1][a:b]), p_(t[2][a:b]), p_(t[3][a:b]),
                                                    p_(t[4][a:b]), p_(t[5][a:b]), p_(t[6][a:b]), b - a, tot,
                                                    p_(self.step_state), float(lr), b1, b2, self.eps, self.weight_decay,
                                                    code, bound, st()), "tn_adamw_multi_bounded")
            if self._pipe is None:
                for _, a, b, tot, code in spans:
                    launch(a, b, tot, code)
            else:
                side, events = self._pipe[2], self._pipe[3]
                side.wait_stream(torch.cuda.current_stream())      # gradients, norm state and the table are ready
                with torch.cuda.stream(side):
                    if self._pipe_delay_cycles:
                        torch.cuda._sleep(int(self._pipe_delay_cycles))
                    for k, (gi, a, b, tot, code) in enumerate(spans):
                        if b > a:
                            launch(a, b, tot, code)
                        if k + 1 == len(spans) or spans[k + 1][0] != gi:
                            events[gi].record(side)
        self._keep = keep                     # tables / gradients stay alive until the next step (async launches)
        return self.norm_sq.sqrt().squeeze(0)

    @property
    def step_count(self) -> int:
        """Number of APPLIED updates (host sync: logging / checkpoint use only)."""
        return int(self.step_state[:1].view(torch.int32).item())

    # ------------------------------------------------------------------ checkpoint
    # The reference's CheckpointManager saves the optimizer through torch.distributed.checkpoint
    # (touchnet/utils/checkpoint.py): DCP flattens nested dicts into string keys and decides per tensor whether it is
    # sharded (a DTensor: every rank saves / reloads ITS shard) or replicated (a plain tensor: saved once, every rank
    # reloads the same bytes).  So the state is keyed by parameter NAME, and under FSDP2 the fp32 master / moments —
    # plain local shards inside this class — are handed out as DTensors with the parameter's mesh and placements.
    def _as_saved(self, i: int, t: torch.Tensor) -> torch.Tensor:
        p = self.params[i]
        mesh = getattr(p, "device_mesh", None)
        if mesh is None:
            return t
        from torch.distributed.tensor import DTensor
        return DTensor.from_local(t, mesh, p.placements, run_check=False, shape=p.shape, stride=p.stride())

    def state_dict(self) -> Dict[str, Any]:
        self.wait_updates()
        return {"step_state": self.step_state.clone(),
                "state": {n: {"master": self._as_saved(i, s["master"]), "exp_avg": self._as_saved(i, s["m"]),
                              "exp_avg_sq": self._as_saved(i, s["v"])}
                          for i, (n, s) in enumerate(zip(self.names, self.state))},
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps,
                          "weight_decay": self.weight_decay, "max_norm": self.max_norm}}

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        st = sd["state"]
        self.wait_updates()
        if len(st) != len(self.state):
            raise ValueError(f"optimizer state has {len(st)} tensors, this optimizer {len(self.state)}")
        self.step_state.copy_(sd["step_state"])
        for i, (n, s) in enumerate(zip(self.names, self.state)):
            src = st[n] if n in st else st[i] if i in st else st[str(i)]     # (pre-round-3 checkpoints: integer keys)
            for key, dst in (("master", s["master"]), ("exp_avg", s["m"]), ("exp_avg_sq", s["v"])):
                t = _local(src[key])
                if t.shape != dst.shape:
                    raise ValueError(f"optimizer state {n}.{key}: shard shape {tuple(t.shape)} != {tuple(dst.shape)}")
                if t.data_ptr() != dst.data_ptr():
                    dst.copy_(t)
            lp = _local(self.params[i].data)
            if lp.dtype == torch.bfloat16:
                lp.copy_(s["master"])
        h = sd.get("hyper", {})
        self.lr = h.get("lr", self.lr)


def update_groups(model: torch.nn.Module, names: List[str]):
    """Groups for `FusedAdamW.pipeline_updates` in the order a forward consumes the parameters `names` (optimizer order):
    [early] + one group per repeated block (the children of the model's outermost `nn.ModuleList`s, in registration
    order) + [late], where `late` are the parameters registered behind the last block (final norm, lm_head) and `early`
    every other parameter outside the blocks.  Returns (groups, blocks) with blocks = [(module, group index)]."""
    strip = lambda n: n.replace("_checkpoint_wrapped_module.", "")
    blocks = []                                                   # (prefix, module)
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.ModuleList) and not any(strip(name).startswith(p + ".") for p, _ in blocks):
            for cname, child in mod.named_children():
                if any(True for _ in child.parameters()):
                    blocks.append((strip(f"{name}.{cname}" if name else cname), child))
    owner = []
    for n in names:
        hit = [bi for bi, (p, _) in enumerate(blocks) if n.startswith(p + ".")]
        owner.append(hit[0] if hit else None)
    inside = [i for i, o in enumerate(owner) if o is not None]
    last = max(inside) if inside else -1
    early = [i for i, o in enumerate(owner) if o is None and i < last]
    late = [i for i, o in enumerate(owner) if o is None and i > last]
    groups, out = [early], []
    for bi, (_, mod) in enumerate(blocks):
        mine = [i for i, o in enumerate(owner) if o == bi]
        if mine:
            out.append((mod, len(groups)))
            groups.append(mine)
    groups.append(late)
    return groups, out


def pipeline_updates_under_forward(model: torch.nn.Module, optimizer: "FusedAdamW"):
    """`optimizer.step()` returns once the norm is known; the parameter updates run on a side stream, block by block, and
    each block's forward waits for ITS update only (forward pre-hooks; the parameters outside the blocks are waited for
    at the model's entry, the ones behind the last block — final norm, lm_head — behind the last block).  The reference
    runs `optimizers.step()` to completion in front of the next forward (touchnet/bin/train.py:467-474); the values
    every kernel reads are the same, only 40 ms of HBM-bound work per step (7 B parameters, one GPU) move under
    MFMA-bound work.  Returns the hook handles."""
    groups, blocks = update_groups(model, optimizer.names)
    optimizer.pipeline_updates(groups)
    late, last_block = len(groups) - 1, (blocks[-1][1] if blocks else None)
    at_entry = [0] if blocks else [0, late]           # groups the model's entry waits for
    fired, learning = [], [True]

    def entry(m, a):
        if learning[0]:
            optimizer.wait_updates()
        for gi in at_entry:
            optimizer.wait_group(gi)

    def block(gi):
        def hook(m, a):
            if learning[0] and gi not in fired:
                fired.append(gi)
            optimizer.wait_group(gi)
        return hook

    def first_forward_done(m, a, o):
        # The FIRST forward (nothing is pending yet) shows which blocks are really entered through __call__ and in which
        # order: a block whose hook never fired (its weights are read by its parent directly) is waited for at the entry
        # from now on, and the launches follow the order of use.
        if not learning[0]:
            return
        learning[0] = False
        silent = [gi for _, gi in blocks if gi not in fired]
        at_entry.extend(silent)
        order = [0] + silent
        for gi in fired:
            order.append(gi)
            if gi == last_block:
                order.append(late)
        if late not in order:
            order.append(late)
            if late not in at_entry:
                at_entry.append(late)
        optimizer.set_launch_order(order)

    handles = [model.register_forward_pre_hook(entry)]
    for mod, gi in blocks:
        handles.append(mod.register_forward_pre_hook(block(gi)))
    if blocks:
        handles.append(blocks[-1][0].register_forward_hook(lambda m, a, o: optimizer.wait_group(late)))
    handles.append(model.register_forward_hook(first_forward_done))
    # anything that reads the parameters outside a forward
    handles.append(model.register_state_dict_pre_hook(lambda m, prefix, keep_vars: optimizer.wait_updates()))
    return handles


def linear_warmup_linear_decay(step: int, warmup: int, total: int, min_ratio: float = 0.0) -> float:
    """WSD-linear LR multiplier (touchnet/utils/optimizer.py:234-322, default 'linear' decay)."""
    if step < warmup:
        return float(step + 1) / float(warmup + 1)
    span = max(1, total - warmup)
    return max(min_ratio, 1.0 - float(step - warmup) / span * (1.0 - min_ratio))


class LRScheduler:
    """What `build_lr_schedulers_fn` returns (the role of LRSchedulersContainer, touchnet/utils/optimizer.py:175-231:
    `step()`, `state_dict()`, `load_state_dict()`): drives `optimizer.lr` with the WSD-linear multiplier."""

    def __init__(self, optimizer: FusedAdamW, base_lr: float, warmup: int, total: int, min_ratio: float = 0.0):
        self.optimizer, self.base_lr, self.warmup, self.total, self.min_ratio = optimizer, base_lr, warmup, total, min_ratio
        self.last_step = 0
        self._apply()

    def _apply(self):
        self.optimizer.lr = self.base_lr * linear_warmup_linear_decay(self.last_step, self.warmup, self.total,
                                                                      self.min_ratio)

    def step(self) -> None:
        self.last_step += 1
        self._apply()

    def get_last_lr(self):
        return [self.optimizer.lr]

    def state_dict(self) -> Dict[str, Any]:
        return {"last_step": self.last_step}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        self.last_step = int(sd["last_step"])
        self._apply()
