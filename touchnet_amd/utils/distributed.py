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
    # TN_DIST_BACKEND=gloo on devices: several ranks may SHARE a GPU (RCCL refuses two ranks on one device, gloo stages
    # device buffers through the host) — how the N > 1 code paths run end to end on a one-GPU box (bench.py, tests)
    backend_env = os.environ.get("TN_DIST_BACKEND")
    if device_type == "cuda":
        if backend_env == "gloo":
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
    force = os.environ.get("TN_FORCE_FSDP") == "1"      # 1-rank RCCL group: exercises the sharded path on one GPU
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this host driver
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")  # distributed.py:391
        # RCCL's own streams on high-priority hardware queues: beside — not behind — the compute queue (utils/zero_dp.py)
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        backend = backend_env or ("nccl" if device_type == "cuda" else "gloo")
        kw = {}
        if device_type == "cuda" and backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, timeout=timedelta(seconds=timeout_s), **kw)
    return rank, local, world


def _gloo_on_devices(device_type: str) -> bool:
    return device_type == "cuda" and os.environ.get("TN_DIST_BACKEND") == "gloo"


class GroupMesh:
    """The slice of torch's DeviceMesh API this path uses — `mesh[name]`, `get_group()`, `size()`, `get_local_rank()`,
    `mesh_dim_names`, `ndim` — over plain gloo groups.  Why not a DeviceMesh: with CUDA present torch builds a mesh's
    groups as "cpu:gloo,cuda:nccl", i.e. device tensors go through RCCL, which refuses two ranks on one GPU; under
    TN_DIST_BACKEND=gloo (several ranks sharing a GPU: bench.py / tests on a one-GPU box) every group must be gloo's.
    Rank layout = DeviceMesh's: row-major over the dimensions in the order given; a flattened name spans its parts."""

    def __init__(self, sizes: dict, flats: dict = None, _views=None, _name=None):
        self._sizes, self._name, self._views = dict(sizes), _name, _views
        if _views is None:
            names = [n for n in sizes]
            world, rank = dist.get_world_size(), dist.get_rank()
            total = 1
            for n in names:
                total *= sizes[n]
            if total != world:
                raise ValueError(f"mesh {sizes} does not cover WORLD_SIZE {world}")
            coords = lambda r: {n: (r // _stride(names, sizes, n)) % sizes[n] for n in names}
            mine = coords(rank)
            self._views = {}
            spans = {n: (n,) for n in names}
            spans.update({k: tuple(p for p in v if p in sizes) for k, v in (flats or {}).items()})
            for view, parts in spans.items():
                if not parts:
                    continue
                # every rank creates every group, in one global order (new_group is collective)
                keys = sorted({tuple(c[n] for n in names if n not in parts) for c in map(coords, range(world))})
                for key in keys:
                    ranks = [r for r in range(world) if tuple(coords(r)[n] for n in names if n not in parts) == key]
                    g = dist.new_group(ranks, backend="gloo")
                    if rank in ranks:
                        local = 0
                        for n in parts:
                            local = local * sizes[n] + mine[n]
                        self._views[view] = (g, len(ranks), local)
        self.mesh_dim_names = tuple(sizes) if _name is None else (_name,)
        self.ndim = len(self.mesh_dim_names)

    def __getitem__(self, names):
        name = names[0] if isinstance(names, (tuple, list)) and len(names) == 1 else names
        if not isinstance(name, str) or name not in self._views:
            raise KeyError(names)
        return GroupMesh(self._sizes, _views=self._views, _name=name)

    def _one(self):
        if self._name is None:
            if len(self._views) and len(self._sizes) == 1:
                return self._views[next(iter(self._sizes))]
            raise RuntimeError("index the mesh by a dimension name first")
        return self._views[self._name]

    def get_group(self, dim=None):
        return self._one()[0]

    def size(self, dim=None):
        return self._one()[1]

    def get_local_rank(self, dim=None):
        return self._one()[2]


def _stride(names, sizes, n):
    s = 1
    for m in names[names.index(n) + 1:]:
        s *= sizes[m]
    return s


def build_dp_mesh(device_type: str, world_size: int):
    """1-D `dp` (= dp_shard) mesh over all ranks: the FSDP2 data-parallel configuration of the reference
    recipes (examples/audio/sft/asr/wenetspeech/run.sh:55-75: dp_shard=8, tp=cp=pp=1)."""
    if _gloo_on_devices(device_type):
        return GroupMesh({"dp": world_size})
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


def dist_min(x: torch.Tensor, group=None) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        x = x.clone()
        dist.all_reduce(x, op=dist.ReduceOp.MIN, group=group)
    return x


def dist_mean(x: torch.Tensor, group=None) -> torch.Tensor:
    """mean over the group's ranks (touchnet/utils/distributed.py:215-217), as a device scalar"""
    n = dist.get_world_size(group) if dist.is_initialized() else 1
    return dist_sum(x, group) / n if n > 1 else x


# ------------------------------------------------------------------------------------------------ parallel dims
from dataclasses import dataclass  # noqa: E402


@dataclass
class ParallelDims:
    """Degrees of each parallelism and the device mesh built from them — field names, validation, mesh-dimension names
    ("pp", "dp_replicate", "dp_shard", "cp", "tp") and the flattened sub-meshes ("dp" for data loading, "dp_shard_cp"
    for parameter sharding, "dp_cp" for the loss all-reduce) are the reference's (touchnet/utils/distributed.py:71-196),
    because `parallelize_fn(model, world_mesh, parallel_dims, job_config)` is called with them (touchnet/bin/train.py:259-261)."""
    dp_replicate: int
    dp_shard: int
    cp: int
    tp: int
    pp: int
    world_size: int
    enable_loss_parallel: bool = False

    def __post_init__(self):
        fixed = self.dp_replicate * self.cp * self.tp * self.pp
        if min(self.dp_replicate, self.cp, self.tp, self.pp) < 1:
            raise AssertionError("Parallelism degree should be >= 1, except for dp_shard")
        if self.dp_shard == -1:
            self.dp_shard = self.world_size // fixed
        if self.dp_shard < 1 or fixed * self.dp_shard != self.world_size:
            raise AssertionError(f"Invalid parallel dims: dp_replicate({self.dp_replicate}) * dp_shard({self.dp_shard}) * "
                                 f"cp({self.cp}) * tp({self.tp}) * pp({self.pp}) != WORLD_SIZE({self.world_size})")

    dp_enabled = property(lambda self: self.dp_replicate > 1 or self.dp_shard > 1)
    dp_replicate_enabled = property(lambda self: self.dp_replicate > 1)
    dp_shard_enabled = property(lambda self: self.dp_shard > 1)
    cp_enabled = property(lambda self: self.cp > 1)
    tp_enabled = property(lambda self: self.tp > 1)
    pp_enabled = property(lambda self: self.pp > 1)
    loss_parallel_enabled = property(lambda self: self.tp > 1 and self.enable_loss_parallel)
    non_data_parallel_size = property(lambda self: self.cp * self.tp * self.pp)

    def build_mesh(self, device_type: str):
        from torch.distributed.device_mesh import init_device_mesh
        degrees = {"pp": self.pp, "dp_replicate": self.dp_replicate, "dp_shard": self.dp_shard, "cp": self.cp,
                   "tp": self.tp}
        names = tuple(n for n, d in degrees.items() if d > 1)
        flat = {"dp": ("dp_replicate", "dp_shard"), "dp_shard_cp": ("dp_shard", "cp"),
                "dp_cp": ("dp_replicate", "dp_shard", "cp")}
        if _gloo_on_devices(device_type):
            return GroupMesh({n: degrees[n] for n in names}, flat)
        mesh = init_device_mesh(device_type, tuple(degrees[n] for n in names), mesh_dim_names=names)
        for flat_name, parts in flat.items():                    # all process groups are created here, up front
            present = tuple(n for n in parts if n in names)
            if present:
                mesh[present]._flatten(mesh_dim_name=flat_name)
        return mesh
