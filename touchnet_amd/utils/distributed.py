"""Process-group / mesh helpers for the data-parallel path (mirrors the parts of
touchnet/utils/distributed.py the hot path uses: init_distributed :349-396, dist_sum :199-220,
the `dp` mesh of ParallelDims.build_mesh :72-196).

One process per GPU; backend "nccl" (= RCCL over xGMI on ROCm) on devices, "gloo" on CPU (tests).
Scalars are reduced as DEVICE tensors — `dist_sum` never calls `.item()` (the reference's does,
distributed.py:204, which stalls the step pipeline once per batch).
"""
from __future__ import annotations

import os
from datetime import timedelta

import torch
import torch.distributed as dist


def init_distributed(device_type: str = None, timeout_s: int = 300) -> tuple[int, int, int]:
    """Returns (rank, local_rank, world_size); a no-op single-process setup when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type is None:
        device_type = "cuda" if torch.cuda.is_available() else "cpu"
    if device_type == "cuda":
        torch.cuda.set_device(local)
    force = os.environ.get("TN_FORCE_FSDP") == "1"      # 1-rank RCCL group: exercises the sharded path on one GPU
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this host driver
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")  # distributed.py:391
        backend = "nccl" if device_type == "cuda" else "gloo"
        kw = {}
        if device_type == "cuda":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, timeout=timedelta(seconds=timeout_s), **kw)
    return rank, local, world


def build_dp_mesh(device_type: str, world_size: int):
    """1-D `dp` (= dp_shard) mesh over all ranks: the FSDP2 data-parallel configuration of the reference
    recipes (examples/audio/sft/asr/wenetspeech/run.sh:55-75: dp_shard=8, tp=cp=pp=1)."""
    from torch.distributed.device_mesh import init_device_mesh
    return init_device_mesh(device_type, (world_size,), mesh_dim_names=("dp",))


def dist_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    """SUM all-reduce of a device scalar; stays on the device (no host sync)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        x = x.clone()
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x


def dist_max(x: torch.Tensor, group=None) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        x = x.clone()
        dist.all_reduce(x, op=dist.ReduceOp.MAX, group=group)
    return x
