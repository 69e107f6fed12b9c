"""Fused clip + AdamW on the HIP kernels (touchnet_amd/csrc/optim.hip).

Hyper-parameters and update rule are the reference's (touchnet/utils/optimizer.py:157-172:
AdamW, betas (0.9, 0.95), weight decay 0.1 on every parameter, eps 1e-8) plus the global-norm clip of
touchnet/utils/distributed.py:426-491 and the skip-on-nonfinite of touchnet/bin/train.py:467-473 — all
evaluated on the device, so the optimizer step issues no host synchronisation.

Precision layout (identical to FSDP2's MixedPrecisionPolicy(param=bf16, reduce=fp32) that the reference
applies, touchnet/models/helper_func.py:165):
  * single GPU:  module parameters are bf16 (what the kernels read); this class owns the fp32 master
                 copy and the fp32 Adam moments, and rewrites the bf16 parameter in the same pass
  * FSDP2:       the sharded parameters ARE the fp32 masters (FSDP all-gathers bf16 copies); the kernel
                 runs on each rank's local shard and the squared norm is all-reduced over the mesh
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch

from touchnet_amd import _C


def _local(t: torch.Tensor) -> torch.Tensor:
    return t._local_tensor if hasattr(t, "_local_tensor") else t


class FusedAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=8e-4, betas=(0.9, 0.95), eps=1e-8,
                 weight_decay=0.1, max_norm: float = 1.0, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.group = process_group
        self.step_count = 0
        dev = _local(self.params[0]).device
        self.state = []
        for p in self.params:
            lp = _local(p.data)
            master = lp if lp.dtype == torch.float32 else lp.detach().float().clone()
            self.state.append(dict(master=master, m=torch.zeros_like(master), v=torch.zeros_like(master)))
        self.norm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.scratch = torch.empty(_C.lib().tn_sumsq_scratch_floats(), dtype=torch.float32, device=dev)
        self._partial, self._keep = None, None

    def _sumsq(self, grads):
        """norm_sq += sum(g^2) over all gradients: one multi-tensor launch per gradient dtype (the gradient
        tensors are new allocations every step, so the pointer table is rebuilt and uploaded each time: ~20 KB)."""
        lib, p_, st = _C.lib(), _C.ptr, _C.stream
        chunk = int(lib.tn_sumsq_multi_chunk())
        by_dtype = {}
        for g in grads:
            if g is not None and g.numel():
                by_dtype.setdefault(g.dtype, []).append(g)
        for dt, gs in by_dtype.items():
            sizes = [g.numel() for g in gs]
            first, tot = [], 0
            for n in sizes:
                first.append(tot)
                tot += (n + chunk - 1) // chunk
            table = torch.tensor([[g.data_ptr() for g in gs], sizes, first], dtype=torch.int64).pin_memory()
            table = table.to(gs[0].device, non_blocking=True)
            if self._partial is None or self._partial.numel() < tot:
                self._partial = torch.empty(tot, dtype=torch.float32, device=gs[0].device)
            _C.check(lib.tn_sumsq_multi(p_(table[0]), p_(table[1]), p_(table[2]), len(gs), tot, p_(self._partial),
                                        p_(self.norm_sq), _C.dcode(gs[0]), st()), "tn_sumsq_multi")
            self._keep = (table, gs)          # alive until the next step (the launch is asynchronous)

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
        self.step_count += 1
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count
        self.norm_sq.zero_()
        grads = [None if p.grad is None else _local(p.grad).contiguous() for p in self.params]
        self._sumsq(grads)
        if self.group is not None:
            torch.distributed.all_reduce(self.norm_sq, group=self.group)
        for p, g, s in zip(self.params, grads, self.state):
            if g is None or not g.numel():
                continue
            lp = _local(p.data)
            shadow = lp if lp.dtype == torch.bfloat16 else None
            _C.check(lib.tn_adamw_step(p_(s["master"]), p_(s["m"]), p_(s["v"]), p_(g), p_(shadow), p_(self.norm_sq),
                                       g.numel(), float(lr), b1, b2, self.eps, self.weight_decay,
                                       float(self.max_norm), bc1, bc2, _C.dcode(g), st()), "tn_adamw_step")
        return self.norm_sq.sqrt().squeeze(0)


def linear_warmup_linear_decay(step: int, warmup: int, total: int, min_ratio: float = 0.0) -> float:
    """WSD-linear LR multiplier (touchnet/utils/optimizer.py:234-322, default 'linear' decay)."""
    if step < warmup:
        return float(step + 1) / float(warmup + 1)
    span = max(1, total - warmup)
    return max(min_ratio, 1.0 - float(step - warmup) / span * (1.0 - min_ratio))
