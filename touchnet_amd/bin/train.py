"""Step driver for the packed path.

The reference's touchnet/bin/train.py cannot run with WORLD_SIZE=1 nor on a host without a GPU
(SURVEY.md §0 fact 5), so the single-GPU configurations need a driver of their own; with more ranks the
same code path runs under FSDP2 exactly as `Trainer.train_step` does (train.py:395-506), minus its host
synchronisations:

  next_batch   train.py:334-393   H2D copy, global `num_sentence` = SUM over dp   (device all-reduce)
  train_step   train.py:395-506   zero_grad -> forward -> loss_fn/acc_fn -> backward -> clip -> AdamW
                                  (NaN/Inf grad norm skips the update — decided on the device)

It consumes a TrainSpec (touchnet_amd.utils.train_spec, the reference's registry surface), so the model
families are switched by `training_model_name` exactly like in the reference.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from touchnet_amd.utils.distributed import ParallelDims, dist_sum
from touchnet_amd.utils.optimizer import FusedAdamW, linear_warmup_linear_decay
from touchnet_amd.utils.train_spec import TrainSpec, get_train_spec


@dataclass
class TrainConfig:
    """The subset of touchnet/bin/__init__.py:65-642 the step needs (same field names)."""
    training_model_name: str = "llama_mi355"
    training_seed: int = 2025
    training_max_norm: float = 1.0
    training_mixed_precision_param: str = "bfloat16"
    training_mixed_precision_reduce: str = "float32"
    training_fsdp_reshard_after_forward: str = "never"
    training_tp_sequence_parallel: bool = True  # TP: norms / residual stream on T/tp rows per rank (the reference's plan)
    training_enable_loss_parallel: bool = False  # TP: vocabulary-parallel lm_head + CE (touchnet/utils/distributed.py:318-323)
    training_dp_engine: str = "flat"           # data parallelism of THIS driver: "flat" = utils/zero_dp.py (flat per-block
                                               # buffers, sharded optimizer state), "fsdp2" = the reference's fully_shard
                                               # (always used under tensor parallelism and behind `parallelize_fn`)
    training_activation_checkpoint_mode: str = "none"             # "none" | "full" | "selective"
    training_activation_checkpoint_selective_ac_option: str = "2"
    training_compile: bool = False
    training_enable_cpu_offload: bool = False
    training_enable_fused_ce: bool = True      # role of `training_enable_liger_kernel`'s fused-linear-CE branch
    training_ce_chunk_tokens: int = 4096       # rows of logits alive at once in the fused lm_head + CE (4096 x V x 2 B)
    training_cp_halo_exchange: bool = True     # CP: point-to-point exchange of the K/V chunks a rank can see (else all-gather)
    training_ce_compact_rows: bool = False     # opt-in: lm_head only on labelled positions (one host sync per step)
    training_pipeline_optimizer: bool = False  # unsharded parameters on a GPU: the AdamW launches run block by block on a
                                               # side stream under the NEXT forward (utils/optimizer.py); same arithmetic.
                                               # Measured: NO gain on the 7B step (profiles/r04e_*: the HBM-bound update
                                               # and the MFMA-bound forward share one power budget), so it is opt-in
    lr_scheduler_lr: float = 8e-4
    lr_scheduler_warmup_steps: int = 2000
    lr_scheduler_steps: int = 100000
    optimizer_weight_decay: float = 0.1


class _MeshView:
    """The slice of a world mesh `parallelize_fn` indexes: `world_mesh["dp_shard_cp"]` / `world_mesh[("dp_shard_cp",)]`."""

    def __init__(self, by_name):
        self.by_name = by_name
        self.ndim = len(by_name)

    def __getitem__(self, names):
        names = (names,) if isinstance(names, str) else tuple(names)
        if len(names) != 1:
            raise KeyError(names)
        return self.by_name[names[0]]


class _ForceShard:
    """ParallelDims view whose `dp_shard_enabled` is True on a 1-rank mesh (TN_FORCE_FSDP=1)."""

    def __init__(self, dims):
        self._d = dims

    def __getattr__(self, k):
        return True if k == "dp_shard_enabled" else getattr(self._d, k)


class Trainer:
    def __init__(self, job: TrainConfig, model_config, device: torch.device, dp_mesh=None,
                 spec: Optional[TrainSpec] = None, optimizer_factory=None, cp_mesh=None, fsdp_mesh=None, tp_mesh=None,
                 cp_emulate=None):
        """`dp_mesh`: 1-D data-parallel mesh (rows are split over it).  `cp_mesh`: 1-D context-parallel mesh (the
        sequence dim is split over it); with CP, parameters are sharded over `fsdp_mesh` = dp x cp flattened
        (the reference's `dp_shard_cp`, touchnet/utils/distributed.py:150-157) and the loss parts of the cp
        ranks add up (train.py:485-494 reduces over `dp_cp`).  `tp_mesh`: 1-D tensor-parallel mesh (or
        models.tensor_parallel.EmulatedTPMesh): the blocks are sharded over it by the spec's `parallelize_fn`
        (touchnet/models/llama/parallelize_llama.py:105-196's slot) before FSDP2 shards the tp-local parameters.
        `cp_emulate = (cp, rank)`: this ONE process plays rank `rank` of a cp-way context-parallel group
        (utils.context_parallel.ContextParallel.emulate; `bench.py --emulate-rank`)."""
        self.job, self.device, self.dp_mesh = job, device, dp_mesh
        self.spec = spec or get_train_spec(job.training_model_name)
        self.dp_group = dp_mesh.get_group() if dp_mesh is not None else None
        self.dp_world = dp_mesh.size() if dp_mesh is not None else 1
        self.cp_group = cp_mesh.get_group() if cp_mesh is not None and cp_mesh.size() > 1 else None
        self.cp_emulate = tuple(cp_emulate) if cp_emulate is not None else None
        self.tp_mesh = tp_mesh if tp_mesh is not None and tp_mesh.size() > 1 else None
        self.cp = None
        if fsdp_mesh is None:
            fsdp_mesh = dp_mesh
        shard_world = fsdp_mesh.size() if fsdp_mesh is not None else 1
        # TN_FORCE_FSDP=1: wrap with FSDP2 even on a 1-rank mesh, so that the sharded path (fully_shard hooks, DTensor
        # parameters, RCCL all-gather / reduce-scatter, the optimizer on local shards) can be exercised on one GPU
        sharded = fsdp_mesh is not None and (shard_world > 1 or os.environ.get("TN_FORCE_FSDP") == "1")
        if (shard_world > 1 or self.cp_group is not None or (self.tp_mesh is not None and not getattr(
                self.tp_mesh, "emulated", False))) and self.device.type == "cuda":
            # Collectives run beside the compute from here on.  The GEMM's persistent form (one workgroup per CU, each
            # walking a FIXED list of tiles) assumes it gets every CU: an RCCL kernel that holds some of them would leave
            # those workgroups — and their whole tile lists — waiting behind it.  One workgroup per tile lets the
            # dispatcher hand tiles to whatever CUs are free; it costs 0-2.5 % per kernel alone on the chip
            # (profiles/r03b_gemm_persistent_asym_sweep.json: p0 vs p1 columns).  An explicit library call, not a change
            # of the process environment (TN_GEMM_PERSIST in the environment still overrides it).
            from touchnet_amd import _C
            _C.lib().tn_gemm_set_persistent(0)
        elif self.device.type == "cuda":
            # (set BOTH ways: the switch is process-wide, and a Trainer without collectives built behind a sharded one —
            #  tests, A/B benches in one process — must not inherit its per-tile launches)
            from touchnet_amd import _C
            _C.lib().tn_gemm_set_persistent(1)
        if self.device.type == "cuda":
            import touchnet_amd.functional as _F
            _F.WGRAD_RETURNS_NEED_SYNC = False      # (the sharding engines built below switch it on for themselves)
        if self.spec.additional_pre_init_fn:
            self.spec.additional_pre_init_fn(job)                      # train.py:121-122
        torch.manual_seed(job.training_seed)
        with torch.device("meta"):                                     # train.py:179-182
            model = self.spec.model_cls(model_config)
        self.model_config = model_config
        self.num_params = self.spec.get_num_params_fn(model)
        self.num_params_wo_emb = self.spec.get_num_params_fn(model, exclude_embedding=True)
        if self.cp_group is not None and not sharded and self.cp_emulate is None:
            # gradients are only reduced across cp ranks by the FSDP reduce-scatter over dp x cp: without it the
            # replicas would silently diverge (the reference always shards over `dp_shard_cp` when cp > 1)
            raise ValueError("context parallelism needs parameter sharding over a mesh that includes the cp ranks: "
                             "pass fsdp_mesh = the flattened dp x cp mesh")
        ac = job.training_activation_checkpoint_mode != "none"
        tp = self.tp_mesh.size() if self.tp_mesh is not None else 1
        engine = os.environ.get("TN_DP_ENGINE", job.training_dp_engine)
        if engine not in ("flat", "fsdp2"):
            raise ValueError(f"training_dp_engine: {engine!r} (flat | fsdp2)")
        # (a test's optimizer stand-in must declare that it takes the engine's flat shards: `takes_shards = True`)
        flat = (sharded and engine == "flat" and tp == 1
                and (optimizer_factory is None or getattr(optimizer_factory, "takes_shards", False)))
        sharded = sharded and not flat                             # from here on: "FSDP2-sharded parameters"
        if sharded or ac or tp > 1:
            # the hook is called the way the reference trainer calls it (train.py:259-261): meta model, the mesh
            # indexed by the reference's dimension names, ParallelDims, job config
            n = fsdp_mesh.size() if sharded else 1
            dims = ParallelDims(dp_replicate=1, dp_shard=n, cp=1, tp=tp, pp=1, world_size=n * tp,
                                enable_loss_parallel=job.training_enable_loss_parallel)
            if sharded and n == 1:                                 # TN_FORCE_FSDP on one rank: still take the FSDP branch
                dims = _ForceShard(dims)
            view = {"dp_shard_cp": fsdp_mesh}
            if tp > 1:
                view["tp"] = self.tp_mesh
            # (the flat engine, when chosen, is this Trainer's own business below: the hook is asked for FSDP2 / TP / AC)
            import dataclasses
            model = self.spec.parallelize_fn(model, _MeshView(view), dims,
                                             dataclasses.replace(job, training_dp_engine="fsdp2"))
        if sharded:                                                # fp32 shards, bf16 compute
            model.to_empty(device=device)
            with torch.no_grad():
                model.post_init()
                if self.spec.additional_post_init_fn:
                    self.spec.additional_post_init_fn(model, device)
        else:
            model.to_empty(device=device)
            with torch.no_grad():
                model.post_init()
                if self.spec.additional_post_init_fn:
                    self.spec.additional_post_init_fn(model, device)
            if device.type == "cuda":
                model.to(getattr(torch, job.training_mixed_precision_param))
        if tp > 1:                                  # shards of one weight must not be drawn identically on the tp ranks
            from touchnet_amd.models.tensor_parallel import reinit_tp_shards
            seq_cfg = getattr(model_config, "text_config", model_config)
            reinit_tp_shards(model, job.training_seed, getattr(seq_cfg, "initializer_range", 0.02))
        self.model = model
        self.dp_engine = None
        if flat:
            from touchnet_amd.utils.zero_dp import FlatShardedDataParallel
            self.dp_engine = FlatShardedDataParallel(model, fsdp_mesh,
                                                     reduce_dtype=getattr(torch, job.training_mixed_precision_reduce))
        if optimizer_factory is not None:          # (CPU tests drive the host logic with a torch optimizer)
            self.optimizer = optimizer_factory(self.dp_engine.named_shards() if flat else model.parameters())
        elif flat:
            self.optimizer = FusedAdamW(self.dp_engine.named_shards(), lr=job.lr_scheduler_lr,
                                        weight_decay=job.optimizer_weight_decay, max_norm=job.training_max_norm,
                                        process_group=fsdp_mesh.get_group())
        else:
            from touchnet_amd.models.tensor_parallel import tp_param_ids
            tp_group, tp_ids = tp_param_ids([model])
            self.optimizer = FusedAdamW(model.named_parameters(), lr=job.lr_scheduler_lr,
                                        weight_decay=job.optimizer_weight_decay, max_norm=job.training_max_norm,
                                        process_group=fsdp_mesh.get_group() if sharded else None,
                                        tp_group=tp_group, tp_param_ids=tp_ids)
            want = os.environ.get("TN_PIPELINE_OPTIMIZER")              # "1" / "0" override the job's flag (A/B runs)
            want = job.training_pipeline_optimizer if want is None else want == "1"
            if (want and device.type == "cuda" and not sharded
                    and not any(hasattr(p, "_local_tensor") for p in model.parameters())):
                from touchnet_amd.utils.optimizer import pipeline_updates_under_forward
                pipeline_updates_under_forward(model, self.optimizer)
        self.step = 0
        # the reference's "dp_cp" mesh (loss / metric all-reduces, train.py:485-494): every rank that holds other rows or
        # another part of the sequence
        self.dp_cp_group = fsdp_mesh.get_group() if fsdp_mesh is not None and shard_world > 1 else None

    # ------------------------------------------------------------------ data
    def next_batch(self, batch: dict) -> dict:
        out = {}
        for k, v in batch.items():
            out[k] = v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v
        ns = out["num_sentence"]
        ns = ns.to(self.device, torch.float32) if isinstance(ns, torch.Tensor) else torch.tensor(
            float(ns), dtype=torch.float32, device=self.device)
        # global over the DATA-parallel ranks (train.py:339-343) — and over those only: the cp / tp peers of a rank hold the
        # same rows (without a dp mesh there is nothing to sum, whatever other groups the process belongs to)
        out["num_sentence"] = dist_sum(ns.reshape(1), self.dp_group) if self.dp_world > 1 else ns.reshape(1)
        if self.cp_group is not None or self.cp_emulate is not None:    # train.py:354-389: shard buffers on dim 1
            from touchnet_amd.utils.context_parallel import ContextParallel
            T = out["labels"].shape[1]
            if self.cp is None or self.cp.T != T:
                self.cp = ContextParallel(self.cp_group, T, emulate=self.cp_emulate)
            if not isinstance(out.get("attention_mask"), torch.Tensor):
                # plain causal rows (the Qwen2-Audio path without packing): one document per row, GLOBAL length
                out["attention_mask"] = torch.ones(out["labels"].shape[0], T, dtype=torch.int64, device=self.device)
            if self.job.training_cp_halo_exchange:                      # document ids: from the HOST copy when there is one
                src = batch.get("attention_mask")
                self.cp.set_documents(src if isinstance(src, torch.Tensor) else out["attention_mask"])
            else:
                self.cp.need = None
            presharded = out.pop("audio_cp_sharded", False)
            if isinstance(out.get("audio_positions"), torch.Tensor) and not presharded:
                # Qwen2-Audio: every rank runs the tower on the clips that touch ITS part of the sequence
                self._shard_audio(batch, out)
            if "labelled_rows_max" in out:
                # the packer's bound counts the whole rows; lm_head + CE run on this rank's part of them
                lab = batch.get("labels")
                if "labelled_rows_max_cp" in out:                       # (the loader counted per cp rank already)
                    out["labelled_rows_max"] = int(out.pop("labelled_rows_max_cp")[self.cp.rank])
                elif isinstance(lab, torch.Tensor) and not lab.is_cuda:
                    out["labelled_rows_max"] = int((self.cp.shard(lab, dim=1) != -100).sum())
                else:
                    out.pop("labelled_rows_max")
            for k in ("input_ids", "labels", "position_ids", "sentence_lens", "input_features", "inputs_embeds"):
                if isinstance(out.get(k), torch.Tensor) and not (k == "input_features" and "audio_rows" in out):
                    out[k] = self.cp.shard(out[k], dim=1)
            out.pop("valid_rows_max", None)                             # (counts the whole rows; cp keeps the padding slots)
            out["context_parallel"] = self.cp                           # attention_mask (doc ids) stays global
        return out

    def _shard_audio(self, batch: dict, out: dict) -> None:
        """host copies of the (small) index tensors decide the split; a loader that hands over device tensors only pays a
        read-back here — bench.py's long-audio workload shards once, ahead of the timed region (`audio_cp_sharded`)"""
        host = lambda k: (batch[k] if isinstance(batch.get(k), torch.Tensor) and not batch[k].is_cuda else out[k].cpu()).numpy()
        feats = out["input_features"]
        rows_per_clip = ((feats.shape[-1] - 1) // 2 + 1) // 2
        clips, pos, rows = self.cp.shard_audio(host("audio_positions"), host("audio_output_lengths"), rows_per_clip)
        dev = lambda a: torch.from_numpy(a).to(self.device, non_blocking=True)
        clips_d = dev(clips)
        out["input_features"] = feats.index_select(0, clips_d)
        out["audio_output_lengths"] = out["audio_output_lengths"].index_select(0, clips_d)
        out["audio_positions"], out["audio_rows"] = dev(pos), dev(rows)

    # ------------------------------------------------------------------ step
    def forward_loss(self, data: dict):
        data = dict(data)
        labels, ns, sl = data.pop("labels"), data.pop("num_sentence"), data.pop("sentence_lens")
        data.pop("shift_labels", None)
        if self.job.training_enable_fused_ce:                       # `pred.loss` branch (train.py:443-445)
            pred = self.model(**data, labels=labels, sentence_lens=sl, num_sentence=ns,
                              ce_chunk_tokens=self.job.training_ce_chunk_tokens,
                              ce_compact=self.job.training_ce_compact_rows)
            return pred.loss, pred.loss_per_token, pred.acc
        pred = self.model(**data)
        loss, per_token = self.spec.loss_fn(pred.logits, labels, sl, ns)
        acc = self.spec.acc_fn(pred.logits, labels) if self.spec.acc_fn else None
        return loss, per_token, acc

    def train_step(self, data: dict) -> dict:
        self.optimizer.zero_grad()
        if self.dp_engine is not None:
            self.dp_engine.zero_grad()
        loss, per_token, acc = self.forward_loss(data)
        # Exactly the reference (train.py:456): backward on the loss normalised by the GLOBAL num_sentence;
        # FSDP2's reduce-scatter then AVERAGES over dp, i.e. gradients are 1/dp of the global-batch mean
        # gradient.  We keep that scale for parity (AdamW is invariant to it, the clip threshold is not).
        loss.backward()
        from touchnet_amd.models.backend import ops as _ops
        if hasattr(_ops(), "sync_wgrad_stream"):
            _ops().sync_wgrad_stream()                   # (weight-gradient GEMMs issued beside the input-gradient chain)
        if self.tp_mesh is not None:
            from touchnet_amd.models.tensor_parallel import reduce_sequence_partial_grads
            reduce_sequence_partial_grads(self.model)    # norm weights saw T/tp rows: sum their gradients over tp
        if self.dp_engine is not None:
            self.dp_engine.finish_backward()     # reduce-scatters ran under the backward; shards go to the optimizer
        lr = self.job.lr_scheduler_lr * linear_warmup_linear_decay(
            self.step, self.job.lr_scheduler_warmup_steps, self.job.lr_scheduler_steps)
        grad_norm = self.optimizer.step(lr)
        if self.dp_engine is not None:
            self.dp_engine.gather_params()       # all-gathers of the updated slices travel under the next forward
        self.step += 1
        return {"loss_per_sample": loss.detach(), "loss_per_token": per_token, "acc": acc, "grad_norm": grad_norm}

    # ------------------------------------------------------------------ metrics / evaluation
    def reduce_metrics(self, loss_per_sample, loss_per_token, acc=None):
        """The five numbers the reference logs (train.py:485-494, 571-583): loss per sample SUMMED over dp x cp (every
        rank's loss is already divided by the global sentence count), loss per token mean / max, accuracy mean / min —
        device scalars, no host sync."""
        from touchnet_amd.utils.distributed import dist_max, dist_mean, dist_min
        g = self.dp_cp_group
        ls, lt = loss_per_sample.detach(), loss_per_token.detach()
        zero = torch.zeros((), device=ls.device)
        if g is None:
            # no dp / cp peers (single process, or tensor parallelism only: the tp ranks hold the SAME full loss, and the
            # helpers would treat group=None as WORLD and sum over them) — the reference reduces over dp_cp only when dp or
            # cp is enabled (train.py:485-494)
            a = acc.detach() if acc is not None else zero
            return (ls, lt, lt, a, a)
        return (dist_sum(ls, g), dist_mean(lt, g), dist_max(lt, g),
                dist_mean(acc.detach(), g) if acc is not None else zero,
                dist_min(acc.detach(), g) if acc is not None else zero)

    @torch.no_grad()
    def dev_step(self, data: dict):
        """train.py:553-586: the training forward under no_grad, reduced metrics"""
        return self.reduce_metrics(*self.forward_loss(data))

    @torch.no_grad()
    def dev(self, batches) -> dict:
        """train.py:588-621 without the logging backend: eval mode, every batch of the dev loader through `dev_step`,
        per-batch metrics averaged (loss, accuracy) / extremes kept (max loss per token, min accuracy)."""
        was_training = self.model.training
        self.model.eval()
        if getattr(self.optimizer, "wait_updates", None):
            self.optimizer.wait_updates()
        try:
            metrics = [self.dev_step(self.next_batch(b)) for b in batches]
        finally:
            self.model.train(was_training)
        if not metrics:
            return {}
        col = lambda i: torch.stack([m[i].reshape(()) for m in metrics])
        return {"global_avg_loss_per_sample": col(0).mean(), "global_avg_loss_per_token": col(1).mean(),
                "global_max_loss_per_token": col(2).max(), "global_avg_acc": col(3).mean(),
                "global_min_acc": col(4).min(), "batches": len(metrics)}
