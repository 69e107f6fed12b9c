"""BASELINE configs D (Qwen2-Audio long audio, context parallel) and E (Kimi-Audio, tensor parallel x FSDP2): the host
logic that splits a batch / a model over the ranks, on CPU (gloo, world size 2 and 4; emulated ranks in one process).
The per-op arithmetic is the oracle's, injected explicitly — the product has no CPU path."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_distributed_cpu import KIMI_TINY, TINY, TorchAdamW, _free_port, _tp_batch, _tp_model

QA_TINY = {"audio_config": {"num_mel_bins": 8, "d_model": 32, "encoder_layers": 2, "encoder_attention_heads": 4,
                            "encoder_ffn_dim": 64, "max_source_positions": 10, "init_std": 0.2},
           "audio_token_index": 15,
           "text_config": dict(TINY, num_hidden_layers=1, tie_word_embeddings=False, model_type="qwen2")}
TA = 5            # tokens per clip of the tiny tower: 20 mel frames -> conv stride 2 -> pool 2


def _qa_batch(T=512, B=1, seed=3):
    from touchnet_amd.data.synthetic import qwen2_audio_long_plan
    tok, n = qwen2_audio_long_plan(15, 15, B, T, seed, tokens_per_clip=TA, resp_per_clip=(1, 3))
    g = torch.Generator().manual_seed(seed)
    tok["input_features"] = torch.randn(n, 8, 20, generator=g)
    return tok, n


# ------------------------------------------------------------------------------------------------ host split
@pytest.mark.parametrize("cp", [2, 4])
def test_shard_audio_partitions_positions_and_rows(cp):
    from touchnet_amd.utils.context_parallel import ContextParallel
    T = 1024
    tok, n = _qa_batch(T, B=2)
    pos, lens = tok["audio_positions"].numpy(), tok["audio_output_lengths"].numpy()
    clip_of = np.repeat(np.arange(n), lens)
    row_of = np.arange(pos.size) - np.repeat(np.cumsum(lens) - lens, lens)
    seen = {}
    straddlers = 0
    for r in range(cp):
        view = ContextParallel(None, T, emulate=(cp, r))
        clips, lpos, rows = view.shard_audio(pos, lens, TA)
        assert len(lpos) == len(rows) and np.all(np.diff(clips) > 0)
        Tc = view.Tc
        b, col = lpos // (2 * Tc), lpos % (2 * Tc)
        gcol = np.where(col < Tc, r * Tc + col, (2 * cp - 1 - r) * Tc + (col - Tc))
        for gp, row in zip(b * T + gcol, rows):
            assert gp not in seen                                      # every position has exactly one owner
            seen[int(gp)] = (int(clips[row // TA]), int(row % TA))
        # the sharded token grid really holds AUDIO placeholders there
        ids = view.shard(tok["input_ids"], 1).reshape(-1)
        assert bool((ids[torch.from_numpy(lpos)] == 15).all())
        assert int((ids == 15).sum()) == len(lpos)
        straddlers += len(clips)
    assert len(seen) == pos.size
    for p, c, rw in zip(pos, clip_of, row_of):
        assert seen[int(p)] == (int(c), int(rw))
    assert straddlers > n                                              # some clips are cut by a chunk boundary: run twice


def test_rank_without_audio_still_runs_the_tower_on_one_clip_and_keeps_no_row():
    from touchnet_amd.utils.context_parallel import ContextParallel
    T = 1024
    view = ContextParallel(None, T, emulate=(2, 1))                    # owns chunks 1 and 2: columns [256, 768)
    pos = np.arange(10, 20)                                            # two clips inside chunk 0
    clips, lpos, rows = view.shard_audio(pos, np.array([5, 5]), TA)
    assert clips.tolist() == [0] and lpos.size == 0 and rows.size == 0


def test_bench_defaults_to_the_baseline_recipe_of_the_workload():
    """`python bench.py --gpus 8 --workload W` without --cp / --tp runs the layout BASELINE.json names
    for W (C: dp 8; D: CP 4 x dp 2; E: TP 2 x dp 4); explicit flags win; one GPU and emulated ranks are never re-shaped."""
    import bench
    assert bench.recipe_degrees("qwen2_audio_7b", 8, None, None) == (1, 1)
    assert bench.recipe_degrees("qwen2_audio_7b_long", 8, None, None) == (4, 1)
    assert bench.parallel_layout(8, *bench.recipe_degrees("qwen2_audio_7b_long", 8, None, None))["dp"] == 2
    assert bench.recipe_degrees("kimi_audio_7b", 8, None, None) == (1, 2)
    assert bench.parallel_layout(8, *bench.recipe_degrees("kimi_audio_7b", 8, None, None))["dp"] == 4
    assert bench.recipe_degrees("qwen2_audio_7b_long", 2, None, None) == (1, 1)        # 2 is not a multiple of 4
    assert bench.recipe_degrees("qwen2_audio_7b_long", 4, None, None) == (4, 1)
    assert bench.recipe_degrees("qwen2_audio_7b_long", 1, None, None) == (1, 1)
    assert bench.recipe_degrees("qwen2_audio_7b_long", 8, 2, None) == (2, 1)           # explicit flag wins
    assert bench.recipe_degrees("kimi_audio_7b", 8, None, 1) == (1, 1)
    assert bench.recipe_degrees("qwen2_audio_7b_long", 1, 4, None, emulate_rank=0) == (4, 1)
    assert bench.recipe_degrees("kimi_audio_7b", 1, None, 2, emulate_rank=1) == (1, 2)


def test_parallel_layout_of_the_bench_flags():
    import bench
    d = bench.parallel_layout(8, cp=4)
    assert (d["dp"], d["cp"], d["tp"], d["emulated"]) == (2, 4, 1, None) and d["label"].startswith("cp4 x dp2")
    e = bench.parallel_layout(8, tp=2)
    assert (e["dp"], e["tp"]) == (4, 2) and e["label"] == "tp2 x dp4"
    assert bench.parallel_layout(1)["label"] == "single-gpu"
    m = bench.parallel_layout(1, cp=4, emulate_rank=3)
    assert m["emulated"] == {"group": "cp", "size": 4, "rank": 3}
    for bad in (dict(world=8, cp=3), dict(world=2, cp=4, emulate_rank=0), dict(world=1, cp=2, tp=2, emulate_rank=0),
                dict(world=1, emulate_rank=0), dict(world=1, tp=2, emulate_rank=2)):
        with pytest.raises(SystemExit):
            bench.parallel_layout(bad.pop("world"), **bad)


def test_bench_launches_its_own_ranks_when_started_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE re-executes under torch.distributed.run on 127.0.0.1 with the same
    arguments (bench.self_launch); under a launcher, or for N = 1, it does nothing; more ranks than GPUs is refused
    unless the gloo switch is on."""
    import subprocess
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setenv("TN_DIST_BACKEND", "gloo")
    assert bench.self_launch(1, ["--gpus", "1"]) is None and not seen
    with pytest.raises(SystemExit) as e:
        bench.self_launch(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("WORLD_SIZE", "8")                      # under a launcher: nothing to do
    assert bench.self_launch(8, ["--gpus", "8"]) is None
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.delenv("TN_DIST_BACKEND")
    with pytest.raises(SystemExit) as e:                       # (no GPU here: 0 devices for 2 ranks)
        bench.self_launch(torch.cuda.device_count() + 2, [])
    assert "GPU(s) visible" in str(e.value.code)


def test_synthetic_plans_of_configs_d_and_e():
    from touchnet_amd.data.synthetic import kimi_audio_plan, qwen2_audio_long_plan
    tok, n = qwen2_audio_long_plan(156032, 151646, 1, 65536, 2025)
    ids, lab, doc = tok["input_ids"][0], tok["labels"][0], tok["attention_mask"][0]
    assert n == int((ids == 151646).sum()) // 750 and n >= 80                      # two 20-minute recordings + filler
    assert torch.equal(tok["audio_positions"], (ids == 151646).nonzero().squeeze(1))
    assert int((lab != -100).sum()) == tok["labelled_rows_max"]
    assert bool((lab[ids == 151646] == -100).all())                                # no label on an audio position
    first = int((doc == 1).sum())
    assert first > 30000 and int(tok["position_ids"][0, first - 1]) == first - 1   # one 40-clip document first
    k = kimi_audio_plan(152064, 152064, 16384, 2, 8192, 2025)
    a, t = k["audio_input_ids"], k["text_input_ids"]
    assert a.shape == t.shape == (2, 8192)
    assert bool(((a >= 152064) <= (t == 0)).all())                                 # audio codes sit on blank text slots
    assert bool((k["labels"][a >= 152064] == -100).all())
    assert int((k["labels"] != -100).sum()) == k["labelled_rows_max"]
    assert int(a.max()) < 168448


# ------------------------------------------------------------------------------------------------ config D on 2 ranks
def _qa_reference(T):
    import oracle.ops as oops
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.qwen2_audio import Qwen2AudioConfig, Qwen2AudioPackedForConditionalGeneration
    torch.manual_seed(13)
    model = Qwen2AudioPackedForConditionalGeneration(Qwen2AudioConfig.from_dict(QA_TINY))
    model.post_init()
    tok, _ = _qa_batch(T)
    with use_ops(oops):
        out = model(**{k: v for k, v in tok.items() if k not in ("num_sentence", "labelled_rows_max")},
                    num_sentence=tok["num_sentence"])
        out.loss.backward()
    return ({k: v.detach().clone() for k, v in model.state_dict().items()},
            {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, float(out.loss), tok)


def _qa_worker(rank, world, port, ref_state, ref_grads, ref_loss, batch, ret, engine="fsdp2"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle.ops as oops
    import touchnet_amd.specs  # noqa: F401
    from torch.distributed.device_mesh import init_device_mesh
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.qwen2_audio import Qwen2AudioConfig
    from touchnet_amd.utils.distributed import init_distributed
    try:
        init_distributed("cpu")
        mesh = init_device_mesh("cpu", (1, world), mesh_dim_names=("dp", "cp"))
        flat = mesh["dp", "cp"]._flatten("dp_cp")
        job = TrainConfig(training_model_name="qwen2_audio_mi355", training_enable_fused_ce=True,
                          training_mixed_precision_param="float32", training_dp_engine=engine)
        factory = (lambda ps: TorchAdamW(ps)) if engine == "fsdp2" else (lambda sh: ShardAdamW(sh, group=flat.get_group()))
        factory.takes_shards = engine == "flat"
        with use_ops(oops):
            tr = Trainer(job, Qwen2AudioConfig.from_dict(QA_TINY), torch.device("cpu"), dp_mesh=mesh["dp"],
                         cp_mesh=mesh["cp"], fsdp_mesh=flat, optimizer_factory=factory)
            assert (tr.dp_engine is not None) == (engine == "flat")
            with torch.no_grad():
                for name, p in tr.model.named_parameters():
                    full = ref_state[name]
                    local = p._local_tensor if hasattr(p, "_local_tensor") else p
                    local.copy_(full.chunk(world, dim=0)[rank] if local.shape != full.shape else full)
            data = tr.next_batch(batch)
            n_all = batch["input_features"].shape[0]
            n_mine = data["input_features"].shape[0]
            assert 0 < n_mine < n_all and data["audio_rows"].shape == data["audio_positions"].shape
            want = int((tr.cp.shard(batch["labels"], 1) != -100).sum())
            assert data["labelled_rows_max"] == want < batch["labelled_rows_max"]
            tr.optimizer.zero_grad()
            loss, _, _ = tr.forward_loss(data)
            total = loss.detach().clone()
            dist.all_reduce(total)
            assert float(total) == pytest.approx(ref_loss, rel=1e-5)
            loss.backward()
            worst = 0.0
            grads = {}
            if engine == "flat":                  # gradients left the parameters: rebuild them from the reduced shards
                tr.dp_engine.finish_backward()
                for b in tr.dp_engine.buckets:
                    full = torch.empty(b.total)
                    dist.all_gather_into_tensor(full, b.shard.grad, group=flat.get_group())
                    for p, o in zip(b.params, b.offsets):
                        grads[id(p)] = full[o:o + p.numel()].view(p.shape)
            for name, p in tr.model.named_parameters():
                if name not in ref_grads:                                # (the tower's frozen sinusoidal positions)
                    assert p.grad is None or not p.requires_grad, name
                    continue
                g = grads[id(p)] if engine == "flat" else p.grad.full_tensor() if hasattr(p.grad, "full_tensor") else p.grad
                worst = max(worst, float((g * world - ref_grads[name]).abs().max()))
            assert worst < 5e-5, worst
        ret[rank] = ("ok", n_mine, n_all)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


@pytest.mark.parametrize("world,T,engine", [(2, 512, "fsdp2"), (4, 1024, "fsdp2"), (2, 512, "flat")])
def test_qwen2_audio_under_context_parallelism_equals_single_process(world, T, engine):
    """Config D's split of a Qwen2-Audio batch: every cp rank runs the audio tower on the clips that touch its part of
    the sequence (clips cut by a chunk boundary on both sides), scatters its rows into its part of the embeddings and
    runs lm_head + CE on its own labelled rows; the loss parts add up to the single-process loss and the FSDP-averaged
    gradients x cp — of the decoder AND of the tower / projector — equal the single-process gradients."""
    ref_state, ref_grads, ref_loss, batch = _qa_reference(T)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_qa_worker, args=(world, _free_port(), ref_state, ref_grads, ref_loss, batch, ret, engine), nprocs=world,
                 join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]
    assert sum(results[r][1] for r in range(world)) > results[0][2]       # straddling clips ran on two ranks


# ------------------------------------------------------------------------------------------------ emulated ranks
def test_emulated_cp_rank_runs_the_real_ranks_shapes_without_a_process_group():
    """`bench.py --cp N --emulate-rank r`: one process, no process group; the batch is split exactly like on the real
    rank r, the K/V buffers have the global length, and the step goes through (values of remote chunks are stand-ins)."""
    import oracle.ops as oops
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.qwen2_audio import Qwen2AudioConfig
    from touchnet_amd.utils.context_parallel import ContextParallel
    assert not dist.is_initialized()
    T, cp = 1024, 4
    batch, n_all = _qa_batch(T)
    job = TrainConfig(training_model_name="qwen2_audio_mi355", training_enable_fused_ce=True,
                      training_mixed_precision_param="float32")
    for r in (0, 3):
        with use_ops(oops):
            tr = Trainer(job, Qwen2AudioConfig.from_dict(QA_TINY), torch.device("cpu"), cp_emulate=(cp, r),
                         optimizer_factory=lambda ps: TorchAdamW(ps))
            data = tr.next_batch(batch)
            view = ContextParallel(None, T, emulate=(cp, r))
            assert torch.equal(data["input_ids"], view.shard(batch["input_ids"], 1))
            clips, pos, rows = view.shard_audio(batch["audio_positions"].numpy(), batch["audio_output_lengths"].numpy(), TA)
            assert torch.equal(data["audio_positions"], torch.from_numpy(pos))
            assert torch.equal(data["input_features"], batch["input_features"][torch.from_numpy(clips)])
            stats = tr.train_step(data)
        assert torch.isfinite(stats["loss_per_sample"]) and torch.isfinite(stats["grad_norm"]).all()
        assert all(p.grad is not None for p in tr.model.audio_tower.parameters() if p.requires_grad)
    assert not dist.is_initialized()


def test_emulated_tp_rank_holds_the_real_ranks_shards():
    import oracle.ops as oops
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.tensor_parallel import EmulatedTPMesh, apply_tp, tp_param_ids
    full = _tp_model()
    ref = {n: p.detach().clone() for n, p in full.named_parameters()}
    for r in range(2):
        m = apply_tp(_tp_model(), EmulatedTPMesh(2, r))
        group, ids = tp_param_ids([m])
        assert getattr(group, "emulated", False) and len(ids) == 4 * 10
        for n, p in m.named_parameters():
            if n in m._tn_tp["sharded_names"]:
                dim = 1 if ("o_proj" in n or "down_proj" in n) else 0
                assert torch.equal(p, ref[n].chunk(2, dim=dim)[r]), n
            else:
                assert torch.equal(p, ref[n]), n
        inputs, labels, sl = _tp_batch()
        with use_ops(oops):
            out = m(**inputs, labels=labels, sentence_lens=sl, num_sentence=4)
            out.loss.backward()
        assert torch.isfinite(out.loss)


def test_tp_shards_are_not_initialised_identically():
    """`post_init` on tp-local shards with one seed would give every tp rank the same numbers."""
    from touchnet_amd.models.kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM
    from touchnet_amd.models.tensor_parallel import EmulatedTPMesh, apply_tp, reinit_tp_shards
    shards = []
    for r in range(2):
        torch.manual_seed(5)
        m = apply_tp(KimiAudioPackedForCausalLM(KimiAudioConfig(**KIMI_TINY)), EmulatedTPMesh(2, r))
        m.post_init()
        reinit_tp_shards(m, seed=5, std=0.02)
        shards.append({n: p.detach().clone() for n, p in m.named_parameters()})
    names = m._tn_tp["sharded_names"]
    for n in shards[0]:
        same = torch.equal(shards[0][n], shards[1][n])
        if n in names and not n.endswith("bias"):
            assert not same, n
            assert 0.01 < float(shards[0][n].std()) < 0.03
        else:
            assert same, n                                             # replicated parameters (and zero biases) agree


def test_tp_shards_and_norm_index_survive_activation_checkpointing():
    """`apply_ac` wraps the blocks AFTER `apply_tp` recorded its parameter names; root-level names then
    carry `_checkpoint_wrapped_module.` and neither the shard re-initialisation nor the tp-aware gradient norm found
    their parameters (identical shards on all tp ranks, clip coefficient without the tp sum)."""
    from types import SimpleNamespace
    from touchnet_amd.models.kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM
    from touchnet_amd.models.parallelize import apply_ac
    from touchnet_amd.models.tensor_parallel import EmulatedTPMesh, apply_tp, reinit_tp_shards, tp_param_ids
    job = SimpleNamespace(training_activation_checkpoint_mode="full", training_activation_checkpoint_selective_ac_option="2")
    shards = []
    for r in range(2):
        torch.manual_seed(5)
        m = apply_tp(KimiAudioPackedForCausalLM(KimiAudioConfig(**KIMI_TINY)), EmulatedTPMesh(2, r))
        n_plain = len(tp_param_ids([m])[1])
        apply_ac(m, job)
        assert any("_checkpoint_wrapped_module." in n for n, _ in m.named_parameters())
        m.post_init()
        reinit_tp_shards(m, seed=5, std=0.02)
        group, ids = tp_param_ids([m])
        assert len(ids) == n_plain and n_plain > 0                       # the optimizer's tp index is complete
        shards.append({n: p.detach().clone() for n, p in m.named_parameters()})
    names = m._tn_tp["sharded_names"]
    checked = 0
    for n in shards[0]:
        plain = n.replace("_checkpoint_wrapped_module.", "")
        if plain in names and not n.endswith("bias"):
            assert not torch.equal(shards[0][n], shards[1][n]), n        # the tp ranks hold DIFFERENT shards
            checked += "_checkpoint_wrapped_module." in n
    assert checked > 0


def test_tp_shard_reinit_under_dim0_sharding_is_not_a_dp_copy():
    """with FSDP2 on top (dp shards along dim 0) every dp rank drew its local rows from the same
    generator state: each tp-local weight came out as dp copies of one row block.  The rows of a tp-local tensor must be
    one draw of its FULL shape, whichever dp rank holds which rows."""
    from torch.distributed.device_mesh import init_device_mesh
    from torch.distributed.tensor import Shard, distribute_tensor
    import torch.distributed as dist
    from touchnet_amd.models.tensor_parallel import reinit_tp_shards
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        mesh = init_device_mesh("cpu", (1,), mesh_dim_names=("dp",))
        lin = torch.nn.Linear(8, 12, bias=False)
        plain = torch.nn.Linear(8, 12, bias=False)
        lin.weight = torch.nn.Parameter(distribute_tensor(lin.weight.detach(), mesh, [Shard(0)]))
        info = {"size": 2, "rank": 1, "sharded_names": {"weight"}, "group": None}
        lin._tn_tp, plain._tn_tp = info, info
        reinit_tp_shards(lin, seed=3, std=0.02)
        reinit_tp_shards(plain, seed=3, std=0.02)
        # one dp rank holds all rows here: the DTensor path (full draw + slice) must reproduce the plain draw
        assert torch.equal(lin.weight._local_tensor, plain.weight)
    finally:
        dist.destroy_process_group()


def test_row_parallel_bias_is_refused():
    from touchnet_amd.models.kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM
    from touchnet_amd.models.tensor_parallel import EmulatedTPMesh, apply_tp
    m = KimiAudioPackedForCausalLM(KimiAudioConfig(**KIMI_TINY))
    m.model.layers[0].self_attn.o_proj.bias = torch.nn.Parameter(torch.zeros(64))
    with pytest.raises(NotImplementedError):
        apply_tp(m, EmulatedTPMesh(2, 0))


# ------------------------------------------------------------------------------------------------ config E on 4 ranks
def _tp_fsdp_batches(dp):
    out = []
    for r in range(dp):
        inputs, labels, sl = _tp_batch()
        g = torch.Generator().manual_seed(40 + r)
        inputs["text_input_ids"] = torch.randint(0, 64, inputs["text_input_ids"].shape, generator=g)
        out.append(dict(inputs, labels=labels, sentence_lens=sl, num_sentence=4))
    return out


def _tp_fsdp_worker(rank, world, port, ref_state, ref_grads, ref_losses, ret, loss_parallel=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle.ops as oops
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.kimi_audio import KimiAudioConfig
    from touchnet_amd.utils.distributed import ParallelDims, init_distributed
    try:
        init_distributed("cpu")
        tp, dp = 2, world // 2
        mesh = ParallelDims(dp_replicate=1, dp_shard=dp, cp=1, tp=tp, pp=1, world_size=world).build_mesh("cpu")
        assert mesh.mesh_dim_names == ("dp_shard", "tp")
        dp_rank, tp_rank = mesh["dp_shard"].get_local_rank(), mesh["tp"].get_local_rank()
        assert (dp_rank, tp_rank) == divmod(rank, tp)                   # tp innermost: neighbours share xGMI links
        job = TrainConfig(training_model_name="kimi_audio_mi355", training_enable_fused_ce=True,
                          training_mixed_precision_param="float32", training_enable_loss_parallel=loss_parallel)
        with use_ops(oops):
            tr = Trainer(job, KimiAudioConfig(**KIMI_TINY), torch.device("cpu"), dp_mesh=mesh["dp"],
                         fsdp_mesh=mesh["dp_shard_cp"], tp_mesh=mesh["tp"], optimizer_factory=lambda ps: TorchAdamW(ps))
            sharded = tr.model._tn_tp["sharded_names"]
            assert len(sharded) == 4 * 10 + (1 if loss_parallel else 0)
            assert tr.model._tn_tp["sequence_parallel"] and len(tr.model._tn_tp["seq_partial_names"]) == 2 * 4 + 2

            def tp_part(name, full):
                if name in sharded:
                    return full.chunk(tp, dim=1 if ("o_proj" in name or "down_proj" in name) else 0)[tp_rank]
                return full
            with torch.no_grad():
                for name, p in tr.model.named_parameters():
                    full = tp_part(name, ref_state[name])
                    local = p._local_tensor
                    assert tuple(p.shape) == tuple(full.shape), (name, p.shape, full.shape)     # DTensor of the tp-local weight
                    local.copy_(full.chunk(dp, dim=0)[dp_rank])
            data = tr.next_batch(_tp_fsdp_batches(dp)[dp_rank])
            assert float(data["num_sentence"]) == 4.0 * dp                # summed over dp only, not over tp
            tr.optimizer.zero_grad()
            loss, _, _ = tr.forward_loss(data)
            assert float(loss) == pytest.approx(ref_losses[dp_rank], rel=1e-5, abs=1e-6)
            loss.backward()
            from touchnet_amd.models.tensor_parallel import reduce_sequence_partial_grads
            reduce_sequence_partial_grads(tr.model)                       # (train_step does this behind the backward)
            worst = 0.0
            for name, p in tr.model.named_parameters():
                if p.grad is None:
                    assert "mimo" in name, name
                    continue
                g = p.grad.full_tensor()
                worst = max(worst, float((g - tp_part(name, ref_grads[name])).abs().max()))
            assert worst < 3e-5, worst
            stats = tr.train_step(data)
            assert torch.isfinite(stats["grad_norm"]).all()
        ret[rank] = ("ok", worst)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def _tp_only_dev_worker(rank, world, port, ref_state, ref_loss, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle.ops as oops
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.kimi_audio import KimiAudioConfig
    from touchnet_amd.utils.distributed import ParallelDims, init_distributed
    try:
        init_distributed("cpu")
        mesh = ParallelDims(dp_replicate=1, dp_shard=1, cp=1, tp=world, pp=1, world_size=world).build_mesh("cpu")
        tp_rank = mesh["tp"].get_local_rank()
        job = TrainConfig(training_model_name="kimi_audio_mi355", training_enable_fused_ce=True,
                          training_mixed_precision_param="float32")
        with use_ops(oops):
            tr = Trainer(job, KimiAudioConfig(**KIMI_TINY), torch.device("cpu"), tp_mesh=mesh["tp"],
                         optimizer_factory=lambda ps: TorchAdamW(ps))
            assert tr.dp_cp_group is None                                  # tp only: nobody to reduce metrics over
            sharded = tr.model._tn_tp["sharded_names"]
            with torch.no_grad():
                for name, p in tr.model.named_parameters():
                    full = ref_state[name]
                    if name in sharded:
                        full = full.chunk(world, dim=1 if ("o_proj" in name or "down_proj" in name) else 0)[tp_rank]
                    p.copy_(full)
            batch = _tp_fsdp_batches(1)[0]
            out = tr.dev([batch, batch])
            # the tp ranks hold the SAME full loss: it must come back as it is, not multiplied by tp
            assert float(out["global_avg_loss_per_sample"]) == pytest.approx(ref_loss, rel=1e-5, abs=1e-6)
            assert out["batches"] == 2 and 0.0 <= float(out["global_avg_acc"]) <= 1.0
            assert float(out["global_max_loss_per_token"]) == pytest.approx(float(out["global_avg_loss_per_token"]), rel=1e-6)
        ret[rank] = ("ok",)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def test_dev_metrics_under_tensor_parallelism_only_are_not_summed_over_the_tp_ranks():
    """`reduce_metrics` handed `dp_cp_group = None` to the reduction helpers, which read None as WORLD —
    a tp-only job summed the identical losses of its tp ranks.  The reference reduces over dp_cp only when dp or cp is
    enabled (touchnet/bin/train.py:485-494)."""
    import oracle.ops as oops
    from touchnet_amd.models.backend import use_ops
    ref = _tp_model()
    with use_ops(oops), torch.no_grad():
        ref.eval()
        loss = float(ref(**_tp_fsdp_batches(1)[0]).loss)
    ref_state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_tp_only_dev_worker, args=(2, _free_port(), ref_state, loss, ret), nprocs=2, join=True)
        results = dict(ret)
    for r in range(2):
        assert results[r][0] == "ok", results[r][1]


@pytest.mark.parametrize("loss_parallel", [False, True])
def test_tensor_parallel_times_fsdp2_on_a_2d_mesh_of_four_ranks(loss_parallel):
    """Config E's layout (TP x FSDP2, here 2 x 2 over gloo) through the Trainer and the reference's ParallelDims mesh:
    tp-local weights are FSDP2-sharded over the dp ranks, the loss of each dp rank equals the single-process one and the
    dp-averaged gradients equal the tp slices of the single-process gradients.  Sequence parallel (the Trainer's default
    under TP), with and without the vocabulary-parallel head."""
    import oracle.ops as oops
    from touchnet_amd.models.backend import use_ops
    ref = _tp_model()
    dp = 2
    losses = []
    with use_ops(oops):
        for b in _tp_fsdp_batches(dp):
            b = dict(b, num_sentence=4 * dp)
            losses.append(ref(**b).loss)
        (sum(losses) / dp).backward()
    ref_state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    ref_grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_tp_fsdp_worker, args=(4, _free_port(), ref_state, ref_grads, [float(l) for l in losses], ret,
                                        loss_parallel), nprocs=4, join=True)
        results = dict(ret)
    for r in range(4):
        assert results[r][0] == "ok", results[r][1]


# ------------------------------------------------------------------------------------------------ flat data-parallel engine
class ShardAdamW:
    """torch.optim.AdamW over the flat engine's shards (CPU stand-in for FusedAdamW: fp32 everywhere here)."""
    takes_shards = True

    def __init__(self, named_shards, lr=1e-2, max_norm=1.0, group=None):
        self.shards = [s for _, s in named_shards]
        self.params = [torch.nn.Parameter(s.data) for s in self.shards]          # share the slices' storage
        self.opt = torch.optim.AdamW(self.params, lr=lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
        self.max_norm, self.group = max_norm, group

    def zero_grad(self):
        self.opt.zero_grad(set_to_none=True)

    def step(self, lr=None):
        sq = torch.zeros(1)
        for p, s in zip(self.params, self.shards):
            p.grad = None if s.grad is None else s.grad.to(p.dtype)
            if p.grad is not None:
                sq += p.grad.float().pow(2).sum()
        dist.all_reduce(sq, group=self.group)
        norm = sq.sqrt()
        coef = torch.clamp(self.max_norm / (norm + 1e-6), max=1.0)
        for p in self.params:
            if p.grad is not None:
                p.grad.mul_(coef)
        self.opt.step()
        return norm.squeeze(0)


def _flat_worker(rank, world, port, model_name, cfg_dict, ref_state, batches, ref_after, ref_norm, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle.ops as oops
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.utils.distributed import build_dp_mesh, init_distributed
    from touchnet_amd.utils.train_spec import get_train_spec
    try:
        init_distributed("cpu")
        mesh = build_dp_mesh("cpu", world)
        job = TrainConfig(training_model_name=model_name, training_enable_fused_ce=True,
                          training_mixed_precision_param="float32", training_dp_engine="flat")
        spec = get_train_spec(model_name)
        cfg = spec.config_cls.from_dict(cfg_dict) if hasattr(spec.config_cls, "from_dict") else spec.config_cls(**cfg_dict)
        factory = lambda shards: ShardAdamW(shards, group=mesh.get_group())
        factory.takes_shards = True
        with use_ops(oops):
            tr = Trainer(job, cfg, torch.device("cpu"), dp_mesh=mesh, optimizer_factory=factory)
            eng = tr.dp_engine
            assert eng is not None and not any(hasattr(p, "_local_tensor") for p in tr.model.parameters())
            with torch.no_grad():
                for name, p in tr.model.named_parameters():
                    p.copy_(ref_state[name])                                  # (views: writes the flat buffers)
            # every trainable parameter is a view into exactly one flat buffer
            owned = sum(p.numel() for b in eng.buckets for p in b.params)
            assert owned == sum(p.numel() for p in tr.model.parameters() if p.requires_grad)
            for b in eng.buckets:
                for p, o in zip(b.params, b.offsets):
                    assert p.data_ptr() == b.flat_p[o:].data_ptr() and o % 128 == 0
                assert b.shard.data.data_ptr() == b.flat_p[rank * b.S:].data_ptr()
            norms = []
            for step in range(2):
                stats = tr.train_step(tr.next_batch(batches[step][rank]))
                norms.append(float(stats["grad_norm"]))
                assert all(p.grad is None for p in tr.model.parameters())     # gradients live in the staging buffers only
            worst = max(float((p.detach() - ref_after[n]).abs().max()) for n, p in tr.model.named_parameters())
            # (AdamW's m / sqrt(v) turns the 1-ulp differences of a mean over 3 ranks into up to ~1e-4 of an lr = 1e-2 step)
            assert worst < (2e-5 if world == 2 else 3e-4), worst
            assert norms[0] == pytest.approx(ref_norm[0], rel=1e-4) and norms[1] == pytest.approx(ref_norm[1], rel=1e-4)
            assert len(eng._pool) <= 3                                        # staging is a small pool, not one per block
        ret[rank] = ("ok", worst, [b.name for b in eng.buckets])
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def _flat_reference(model_cls, cfg, batches, world, fwd):
    """single process: two AdamW steps on the dp-averaged gradient of the per-rank losses"""
    import oracle.ops as oops
    from touchnet_amd.models.backend import use_ops
    torch.manual_seed(17)
    model = model_cls(cfg)
    model.post_init()
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    norms = []
    with use_ops(oops):
        for step in range(2):
            opt.zero_grad(set_to_none=True)
            total_ns = sum(b["num_sentence"] for b in batches[step])
            loss = sum(fwd(model, b, total_ns).loss for b in batches[step]) / world
            loss.backward()
            norms.append(float(torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0)))
            opt.step()
    return state, {n: p.detach().clone() for n, p in model.named_parameters()}, norms


@pytest.mark.parametrize("world", [2, 3])
def test_flat_data_parallel_engine_trains_like_one_process(world):
    """utils/zero_dp.py on gloo: parameters as views of flat per-block buffers, gradients cast-copied into pooled staging
    buffers by the post-accumulate hooks, reduce-scatter (AVG) per block, AdamW on each rank's slice, in-place all-gather
    — two optimizer steps end at the same parameters and report the same global gradient norm as one process on the
    averaged loss.  World size 3: padded buckets (shard boundaries inside parameters)."""
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    cfg_dict = dict(TINY, num_hidden_layers=3, tie_word_embeddings=False)
    batches = [[text_batch(16, 2, 32, seed=100 + 10 * s + r, max_len=9) for r in range(world)] for s in range(2)]
    fwd = lambda m, b, ns: m(input_ids=b["input_ids"], position_ids=b["position_ids"], attention_mask=b["attention_mask"],
                             labels=b["labels"], sentence_lens=b["sentence_lens"], num_sentence=ns)
    state, after, norms = _flat_reference(PackedCausalLM, DecoderConfig.from_dict(cfg_dict), batches, world, fwd)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_flat_worker, args=(world, _free_port(), "llama_mi355", cfg_dict, state, batches, after, norms, ret),
                 nprocs=world, join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]
    assert results[0][2] == ["block.0.0", "block.0.1", "block.0.2", "rest.model.embed_tokens", "rest.model.norm",
                             "rest.lm_head"]


def _reference_hooks_worker(rank, world, port, cfg_dict, ref_state, batches, ref_after, ref_norm, ret, rep=1,
                            reference_job=False):
    """The flat engine driven ONLY through the TrainSpec hooks, in the order touchnet/bin/train.py calls them
    (tests/golden/boundary.json: `model_setup_sequence` :259-297, `train_step_sequence` :396-474)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import json
    import oracle.ops as oops
    import touchnet_amd.specs as specs
    from touchnet_amd.bin.train import TrainConfig, _MeshView
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.utils.distributed import ParallelDims, build_dp_mesh, init_distributed
    from touchnet_amd.utils.train_spec import get_train_spec
    from touchnet_amd.utils.zero_dp import FlatEngineOptimizer
    try:
        init_distributed("cpu")
        mesh = build_dp_mesh("cpu", world)
        fixture = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "boundary.json")))
        specs.OPTIMIZER_FACTORY = lambda shards, group: ShardAdamW(shards, group=group)
        spec = get_train_spec("llama_mi355")
        job = TrainConfig(training_model_name="llama_mi355", training_enable_fused_ce=False,
                          training_mixed_precision_param="float32", training_dp_engine="flat")
        if reference_job:
            # the job config of an UNCHANGED reference run: exactly the reference TrainConfig's fields with their defaults
            # (tests/golden/boundary.json `train_config_fields`, touchnet/bin/__init__.py:65-642) — no engine field exists,
            # the flat engine is selected by TN_DP_ENGINE (models/parallelize.py)
            import types
            fields = dict(fixture["train_config_fields"])
            assert "training_dp_engine" not in fields
            fields.update(training_model_name="llama_mi355", training_mixed_precision_param="float32",
                          training_activation_checkpoint_mode="none", lr_scheduler_lr=job.lr_scheduler_lr,
                          lr_scheduler_warmup_steps=job.lr_scheduler_warmup_steps,
                          lr_scheduler_steps=job.lr_scheduler_steps, optimizer_weight_decay=job.optimizer_weight_decay,
                          training_max_norm=job.training_max_norm)
            job = types.SimpleNamespace(**fields)
            os.environ["TN_DP_ENGINE"] = "flat"
        dims = ParallelDims(dp_replicate=rep, dp_shard=world // rep, cp=1, tp=1, pp=1, world_size=world,
                            enable_loss_parallel=False)
        # HSDP (rep > 1): the reference's 2-D world mesh (touchnet/utils/distributed.py:139-157) with its flattened views
        world_mesh = dims.build_mesh("cpu") if rep > 1 else _MeshView({"dp_shard_cp": mesh})
        dev = torch.device("cpu")
        st = {}
        torch.manual_seed(17)
        with torch.device("meta"):                                       # train.py:179-182
            st["model"] = spec.model_cls(spec.config_cls.from_dict(cfg_dict))
        setup = {
            "self.train_spec.parallelize_fn": lambda: st.__setitem__("model", spec.parallelize_fn(
                st["model"], world_mesh, dims, job)),
            "model.to_empty": lambda: st["model"].to_empty(device=dev),
            "model.post_init": lambda: st["model"].post_init(),
            "self.train_spec.additional_post_init_fn": lambda: spec.additional_post_init_fn(st["model"], dev),
            "model.train": lambda: st["model"].train(),
            "model.to": lambda: st.__setitem__("model", st["model"].to(torch.float32)),
            "self.train_spec.build_optimizers_fn": lambda: st.__setitem__("opt", spec.build_optimizers_fn([st["model"]], job)),
            "self.train_spec.build_lr_schedulers_fn": lambda: st.__setitem__("lrs", spec.build_lr_schedulers_fn(st["opt"], job)),
        }
        first = next(line for name, line in fixture["model_setup_sequence"] if name.endswith("parallelize_fn"))
        with use_ops(oops), torch.no_grad():
            for name, line in fixture["model_setup_sequence"]:
                if line >= first:                                        # (earlier entries: the pipeline-parallel branch)
                    setup[name]()
                    if name.endswith("parallelize_fn"):                  # still on the meta device, nothing sharded yet
                        assert all(p.is_meta for p in st["model"].parameters()) and st["model"]._tn_flat_dp is not None
        model, opt, lrs = st["model"], st["opt"], st["lrs"]
        assert isinstance(opt, FlatEngineOptimizer) and model._tn_flat_engine is opt.engine
        assert not any(hasattr(p, "_local_tensor") for p in model.parameters())      # no FSDP2 DTensors
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(ref_state[name])                                 # (views: writes the flat buffers)
        opt.inner.opt.param_groups[0]["lr"] = 1e-2                       # (the reference run's constant rate)
        norms = []
        with use_ops(oops):
            for step in range(2):
                b = dict(batches[step][rank])
                ns = torch.tensor([float(sum(x["num_sentence"] for x in batches[step]))])      # train.py:339-343
                labels, sl = b.pop("labels"), b.pop("sentence_lens")
                b.pop("num_sentence")
                data = {k: b[k] for k in ("input_ids", "position_ids", "attention_mask")}
                seen = {"zero": 0}
                for name, line in fixture["train_step_sequence"]:
                    if name.endswith("optimizers.zero_grad"):
                        seen["zero"] += 1
                        if seen["zero"] == 1:
                            opt.zero_grad()
                            pred = model(**data)                          # train.py:436
                        elif not torch.isfinite(norm):                    # train.py:467-471: the skip branch
                            opt.zero_grad()
                    elif name.endswith("loss_fn"):
                        loss, _ = spec.loss_fn(pred.logits, labels, sl, ns)
                    elif name.endswith("acc_fn"):
                        spec.acc_fn(pred.logits, labels)
                    elif name.endswith(".backward"):
                        loss.backward()
                    elif name == "clip_grad_norm_":
                        # the reference's clip sees no gradient at all: norm 0, nothing scaled, no nan
                        assert all(p.grad is None for p in model.parameters())
                        norm = torch.nn.utils.clip_grad_norm_([p for p in model.parameters()], job.training_max_norm)
                        assert float(norm) == 0.0
                    elif name.endswith("optimizers.step"):
                        if torch.isfinite(norm):
                            opt.step()
                    elif name.endswith("lr_schedulers.step"):
                        lrs.step()
                        opt.inner.opt.param_groups[0]["lr"] = 1e-2
                norms.append(float(opt.last_grad_norm))
        worst = max(float((p.detach() - ref_after[n]).abs().max()) for n, p in model.named_parameters())
        assert worst < (2e-5 if world == 2 else 3e-4), worst
        assert norms[0] == pytest.approx(ref_norm[0], rel=1e-4) and norms[1] == pytest.approx(ref_norm[1], rel=1e-4)
        if rep > 1:
            assert opt.engine.rep_group is not None and opt.engine.world == world // rep
        ret[rank] = ("ok", worst)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def test_flat_engine_behind_the_reference_hooks_in_the_reference_call_order():
    """`parallelize_fn` (meta model) -> to_empty / post_init / .to(float32) -> `build_optimizers_fn` ->
    per step zero_grad / backward / clip_grad_norm_ / step, replayed from the call-order fixture on 2 gloo ranks, ends two
    AdamW steps at the same parameters and gradient norms as one process on the averaged loss (= what the repo's own
    Trainer reaches with the same engine, test_flat_data_parallel_engine_trains_like_one_process)."""
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    world = 2
    cfg_dict = dict(TINY, num_hidden_layers=3, tie_word_embeddings=False)
    batches = [[text_batch(16, 2, 32, seed=100 + 10 * s + r, max_len=9) for r in range(world)] for s in range(2)]
    fwd = lambda m, b, ns: m(input_ids=b["input_ids"], position_ids=b["position_ids"], attention_mask=b["attention_mask"],
                             labels=b["labels"], sentence_lens=b["sentence_lens"], num_sentence=ns)
    state, after, norms = _flat_reference(PackedCausalLM, DecoderConfig.from_dict(cfg_dict), batches, world, fwd)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_reference_hooks_worker, args=(world, _free_port(), cfg_dict, state, batches, after, norms, ret),
                 nprocs=world, join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]


def test_unchanged_reference_job_selects_the_flat_engine_through_the_environment():
    """the reference's TrainConfig has no `training_dp_engine` and its argument parser rejects unknown
    flags; with TN_DP_ENGINE=flat in the environment `parallelize_fn` picks the flat engine for a job config that holds the
    reference's field set ONLY, and the hooks — replayed in the reference's call order — train like one process."""
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    world = 2
    cfg_dict = dict(TINY, num_hidden_layers=2, tie_word_embeddings=False)
    batches = [[text_batch(16, 2, 32, seed=300 + 10 * s + r, max_len=9) for r in range(world)] for s in range(2)]
    fwd = lambda m, b, ns: m(input_ids=b["input_ids"], position_ids=b["position_ids"], attention_mask=b["attention_mask"],
                             labels=b["labels"], sentence_lens=b["sentence_lens"], num_sentence=ns)
    state, after, norms = _flat_reference(PackedCausalLM, DecoderConfig.from_dict(cfg_dict), batches, world, fwd)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_reference_hooks_worker, args=(world, _free_port(), cfg_dict, state, batches, after, norms, ret, 1, True),
                 nprocs=world, join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]


def test_parallelize_fn_defaults_to_fsdp2_without_the_switch(monkeypatch):
    """no attribute and no TN_DP_ENGINE: the reference's behaviour (FSDP2); a bad value is refused loudly"""
    import types
    from touchnet_amd.models import parallelize as P
    seen = {}
    monkeypatch.setattr(P, "apply_fsdp", lambda model, mesh, **kw: seen.setdefault("fsdp", mesh))
    dims = types.SimpleNamespace(pp_enabled=False, tp_enabled=False, dp_shard_enabled=True, cp_enabled=False,
                                 dp_replicate_enabled=False, loss_parallel_enabled=False)
    job = types.SimpleNamespace(training_activation_checkpoint_mode="none")
    monkeypatch.delenv("TN_DP_ENGINE", raising=False)
    P.parallelize_packed(torch.nn.Linear(2, 2), {("dp_shard_cp",): "mesh"}, dims, job)
    assert seen == {"fsdp": "mesh"}
    monkeypatch.setenv("TN_DP_ENGINE", "zero9")
    with pytest.raises(ValueError):
        P.parallelize_packed(torch.nn.Linear(2, 2), {("dp_shard_cp",): "mesh"}, dims, job)


def test_flat_engine_hsdp_two_replicas_of_two_shards_behind_the_hooks():
    """HSDP on the flat engine (dp_replicate = 2 x dp_shard = 2 on 4 gloo ranks, the reference's 2-D mesh): the state is
    sharded over dp_shard, the reduced gradient shards are averaged over dp_replicate behind each reduce-scatter — four
    data-parallel ranks, each with its own batch, end two AdamW steps where one process on the averaged loss ends."""
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    world, rep = 4, 2
    cfg_dict = dict(TINY, num_hidden_layers=2, tie_word_embeddings=False)
    batches = [[text_batch(16, 2, 32, seed=200 + 10 * s + r, max_len=9) for r in range(world)] for s in range(2)]
    fwd = lambda m, b, ns: m(input_ids=b["input_ids"], position_ids=b["position_ids"], attention_mask=b["attention_mask"],
                             labels=b["labels"], sentence_lens=b["sentence_lens"], num_sentence=ns)
    state, after, norms = _flat_reference(PackedCausalLM, DecoderConfig.from_dict(cfg_dict), batches, world, fwd)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_reference_hooks_worker, args=(world, _free_port(), cfg_dict, state, batches, after, norms, ret, rep),
                 nprocs=world, join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]


def test_flat_engine_with_a_branch_that_takes_no_part_in_the_step():
    """Kimi-Audio's text-head step never runs the mimo layers: their buckets receive no gradient on any rank, are skipped
    by the engine and the optimizer alike (no collective is issued for them) and keep their parameters."""
    world = 2
    batches = [[dict(_tp_fsdp_batches(world)[r]) for r in range(world)] for _ in range(2)]
    from touchnet_amd.models.kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM
    fwd = lambda m, b, ns: m(**{k: v for k, v in b.items() if k != "num_sentence"}, num_sentence=ns)
    state, after, norms = _flat_reference(KimiAudioPackedForCausalLM, KimiAudioConfig(**KIMI_TINY), batches, world, fwd)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_flat_worker, args=(world, _free_port(), "kimi_audio_mi355", KIMI_TINY, state, batches, after, norms,
                                     ret), nprocs=world, join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]
    assert "block.1.0" in results[0][2]                                       # the mimo block has a bucket of its own


def test_split_k_plan_of_the_hand_written_gemm():
    """functional.split_k: parts only for outputs of at most half the CUs' worth of tiles with >= 16 stages; tiles x parts
    <= 256; >= 8 stages per part; contraction-contiguous operands need an even split."""
    import touchnet_amd.functional as F
    assert F.split_k(1280, 1280, 30000, True, True) == 10            # 25 tiles: 250 units of 47 stages
    assert F.split_k(1280, 5120, 30000, True, True) == 2
    assert F.split_k(4096, 4096, 16384, True, True) == 1             # 256 tiles already
    assert F.split_k(1280, 1280, 512, True, True) == 1               # too shallow
    assert F.split_k(1024, 512, 4096, False, False) == 8             # 8 tiles, 64 stages: min(32, 8) parts, even
    assert F.split_k(1024, 512, 4096 + 64 * 3, False, False) == 1    # 67 stages (prime): no even split
    assert F.split_k(1000, 776, 2048, False, True) == 4


@pytest.mark.parametrize("n,T,C,O,stride", [(2, 20, 8, 16, 1), (3, 20, 8, 64, 2), (1, 7, 5, 8, 2)])
def test_conv_stem_im2col_view_scheme_on_cpu(n, T, C, O, stride, monkeypatch):
    """The HOST logic of functional._Conv1dK3 (zero-separated clip slots, overlapping-row im2col views, tap order of the
    reshaped weight, overlap-add input gradient) with the GEMM replaced by torch matmuls: equals torch's conv1d, forward and
    all gradients.  (The kernel itself is held to torch on the device, tests/test_kernels_gpu.py.)"""
    import touchnet_amd.functional as F

    def gemm(segs, a_kmaj=False, b_kmaj=False, bias=None, out=None, accumulate=False, out_t=None):
        (a, b), = segs
        a2 = a.t() if a_kmaj else a
        b2 = b if b_kmaj else b.t()
        r = a2.double() @ b2.double()
        return (r + bias.double() if bias is not None else r).to(a.dtype)
    monkeypatch.setattr(F, "gemm", gemm)
    monkeypatch.setattr(F, "column_sum", lambda t: t.sum(0))
    g = torch.Generator().manual_seed(n + T + C)
    x = torch.randn(n, T, C, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(O, C, 3, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(O, generator=g, dtype=torch.float64, requires_grad=True)
    ref = torch.nn.functional.conv1d(x.transpose(1, 2), w, b, stride=stride, padding=1).transpose(1, 2)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    gx, gw, gb = torch.autograd.grad(ref, (x, w, b), dy)
    full = F._Conv1dK3.apply(x, w, b, stride, True)
    t_out = (T + 2 - 3) // stride + 1
    got = full[:, :t_out]
    assert got.shape == ref.shape and torch.allclose(got, ref, atol=1e-10)
    hx, hw, hb = torch.autograd.grad(got, (x, w, b), dy)
    assert torch.allclose(hx, gx, atol=1e-5) and torch.allclose(hw, gw, atol=1e-10) and torch.allclose(hb, gb, atol=1e-10)
