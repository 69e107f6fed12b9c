"""Tensor and context parallelism THROUGH the HIP kernels on one MI355X (BASELINE configs D / E; world size > 1 is
covered over gloo on CPU in tests/test_distributed_cpu.py and tests/test_sharded_workloads_cpu.py, the 8-GPU run is the
driver's):
  * the tp ranks' parts — each computed by the HIP kernels on that rank's shards — add up to the unsharded block, forward and
    backward (the sum IS the all-reduce the real group performs);
  * `apply_tp` and the K/V halo exchange over a 1-rank RCCL group (the collectives really are issued on the device and the
    HIP autograd nodes run between them) train like the plain model;
  * an emulated cp rank (bench.py --cp 4 --emulate-rank r) runs a whole Qwen2-Audio step through the split attention."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

DEV = "cuda"
CFG = dict(model_type="qwen2", hidden_size=512, intermediate_size=1536, num_attention_heads=8, num_hidden_layers=2,
           num_key_value_heads=4, head_dim=64, vocab_size=1024, tie_word_embeddings=False, rope_theta=1e6,
           rms_norm_eps=1e-6, initializer_range=0.05)


@pytest.fixture(scope="module")
def rccl_single_rank():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    yield
    if created:
        dist.destroy_process_group()


def _packed_batch(B, T, vocab, seed=0):
    from touchnet_amd.data.synthetic import text_batch
    b = text_batch(vocab, B, T, seed=seed, min_len=40, max_len=300)
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}


def _model(cfg_dict=CFG, seed=0):
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    torch.manual_seed(seed)
    m = PackedCausalLM(DecoderConfig.from_dict(cfg_dict))
    m.post_init()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.normal_(std=0.05)
    return m.to(DEV, torch.bfloat16)


def _rel(a, b):
    return float((a.float() - b.float()).abs().max()) / float(b.float().abs().max().clamp_min(1e-6))


def test_tp_rank_parts_through_hip_kernels_add_up_to_the_unsharded_block():
    from touchnet_amd.models.tensor_parallel import EmulatedTPMesh, apply_tp
    import touchnet_amd.functional as F
    full = _model()
    state = {k: v.clone() for k, v in full.state_dict().items()}
    B, T, tp = 2, 1024, 2
    batch = _packed_batch(B, T, 1024)
    blk = full.model.layers[0]
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.randn(B, T, 512, device=DEV, generator=g) * 0.5).bfloat16().requires_grad_()
    cos, sin = full.model.rotary_emb(batch["position_ids"], torch.bfloat16)
    mask = F.build_packed_mask(batch["attention_mask"])
    da = (torch.randn(B, T, 512, device=DEV, generator=g) * 0.1).bfloat16()
    dm = (torch.randn(B, T, 512, device=DEV, generator=g) * 0.1).bfloat16()

    def run(block, xin):
        a = block.self_attn(xin, cos, sin, mask)
        m = block.mlp(xin)
        torch.autograd.backward([a, m], [da, dm])
        return a.detach(), m.detach()

    a_ref, m_ref = run(blk, x)
    dx_ref = x.grad.clone()
    ref_grads = {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None}
    a_sum = m_sum = dx_sum = 0
    for r in range(tp):
        part = _model()
        part.load_state_dict(state)
        apply_tp(part, EmulatedTPMesh(tp, r))
        pb = part.model.layers[0]
        assert pb.self_attn.num_heads == 4 and pb.self_attn.num_kv_heads == 2 and pb.mlp.gate_proj.weight.shape == (768, 512)
        xr = x.detach().clone().requires_grad_()
        a, m = run(pb, xr)
        a_sum, m_sum, dx_sum = a_sum + a.float(), m_sum + m.float(), dx_sum + xr.grad.float()
        for n, p in pb.named_parameters():
            if p.grad is None:
                continue
            want = ref_grads[n]
            if p.shape != want.shape:
                want = want.chunk(tp, dim=1 if ("o_proj" in n or "down_proj" in n) else 0)[r]
                assert _rel(p.grad, want) < 3e-2, (r, n, _rel(p.grad, want))
    # the sums are what the forward / backward all-reduces of a real tp group deliver (fp32 here, bf16 ring there)
    assert _rel(a_sum, a_ref) < 2e-2 and _rel(m_sum, m_ref) < 2e-2, (_rel(a_sum, a_ref), _rel(m_sum, m_ref))
    assert _rel(dx_sum, dx_ref) < 2e-2, _rel(dx_sum, dx_ref)


def _loss_and_grads(model, batch, **kw):
    model.zero_grad()
    out = model(input_ids=batch["input_ids"], position_ids=batch["position_ids"], attention_mask=batch["attention_mask"],
                labels=batch["labels"], sentence_lens=batch["sentence_lens"], num_sentence=batch["num_sentence"], **kw)
    out.loss.backward()
    return float(out.loss), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}


def test_apply_tp_over_a_single_rank_rccl_group_is_the_plain_model(rccl_single_rank):
    from torch.distributed.device_mesh import init_device_mesh
    from touchnet_amd.models.tensor_parallel import apply_tp, tp_param_ids
    batch = _packed_batch(2, 1024, 1024, seed=2)
    plain = _model(seed=3)
    loss0, g0 = _loss_and_grads(plain, batch)
    wrapped = _model(seed=3)
    mesh = init_device_mesh("cuda", (1,), mesh_dim_names=("tp",))
    apply_tp(wrapped, mesh["tp"])                         # all-reduces over RCCL around the attention and MLP nodes
    group, ids = tp_param_ids([wrapped])
    assert group is not None and len(ids) == 2 * 10
    loss1, g1 = _loss_and_grads(wrapped, batch)
    assert loss1 == pytest.approx(loss0, rel=1e-6)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n


@pytest.mark.parametrize("B", [1, 2])
def test_kv_halo_exchange_path_over_a_single_rank_rccl_group_matches_the_plain_attention(rccl_single_rank, B):
    """cp = 1 over RCCL: the rank owns both chunks, so `_forward_context_parallel` (K/V first, exchange issued on the side
    stream, query path, own-chunk attention, LSE merge with an empty remote part, halo return in backward) must reproduce
    the plain packed attention path."""
    from touchnet_amd.utils.context_parallel import ContextParallel
    T = 2048
    batch = _packed_batch(B, T, 1024, seed=4)
    model = _model(seed=5)
    loss0, g0 = _loss_and_grads(model, batch)
    cp = ContextParallel(dist.group.WORLD, T)
    assert (cp.cp, cp.rank, cp.Tc) == (1, 0, T // 2)
    cp.set_documents(batch["attention_mask"].cpu())
    loss1, g1 = _loss_and_grads(model, batch, context_parallel=cp)
    assert abs(loss1 - loss0) / abs(loss0) < 2e-3, (loss0, loss1)
    for n in g0:
        assert _rel(g1[n], g0[n]) < 4e-2, (n, _rel(g1[n], g0[n]))


@pytest.mark.parametrize("rank", [0, 3])
def test_emulated_cp_rank_steps_qwen2_audio_through_the_split_attention(rank):
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.data.synthetic import qwen2_audio_long_plan
    from touchnet_amd.models.qwen2_audio import Qwen2AudioConfig
    cfg = Qwen2AudioConfig.from_dict({
        "audio_config": {"d_model": 128, "encoder_attention_heads": 2, "encoder_ffn_dim": 256, "encoder_layers": 2,
                         "max_source_positions": 50, "num_mel_bins": 16},
        "audio_token_index": 1000, "text_config": dict(CFG, vocab_size=1024)})
    T, cp, ta = 4096, 4, 25                                        # 100 mel frames -> 50 -> 25 tokens per clip
    tok, n = qwen2_audio_long_plan(1000, 1000, 1, T, 7, tokens_per_clip=ta, resp_per_clip=(2, 5))
    g = torch.Generator().manual_seed(0)
    tok["input_features"] = torch.randn(n, 16, 100, generator=g)
    job = TrainConfig(training_model_name="qwen2_audio_mi355", training_enable_fused_ce=True, lr_scheduler_lr=1e-3,
                      lr_scheduler_warmup_steps=0)
    tr = Trainer(job, cfg, torch.device(DEV), cp_emulate=(cp, rank))
    data = tr.next_batch(tok)
    assert data["input_ids"].shape == (1, T // cp) and data["attention_mask"].shape == (1, T)
    assert 0 < data["input_features"].shape[0] < n
    assert data["labelled_rows_max"] == int((tr.cp.shard(tok["labels"], 1) != -100).sum())
    losses = [float(tr.train_step(tr.next_batch(tok))["loss_per_sample"]) for _ in range(4)]
    assert all(l == l and l > 0 for l in losses) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("reduce", ["float32", "bfloat16"])
def test_flat_data_parallel_engine_over_a_single_rank_rccl_group_trains_like_the_plain_model(rccl_single_rank, reduce, monkeypatch):
    """utils/zero_dp.py on the device: parameters re-pointed into flat buffers (the HIP kernels read the views), gradients
    cast-copied into the staging pool by the hooks, reduce_scatter_tensor / in-place all_gather_into_tensor over RCCL on the
    side stream, FusedAdamW on the flat slices — three steps give the unsharded trainer's losses."""
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig
    from touchnet_amd.utils.distributed import build_dp_mesh
    cfg = DecoderConfig.from_dict(dict(CFG, model_type="llama"))
    job = dict(training_model_name="llama_mi355", training_enable_fused_ce=True, lr_scheduler_warmup_steps=0,
               lr_scheduler_lr=1e-3, training_mixed_precision_reduce=reduce)
    batches = [text_batch(1024, 4, 512, seed=s, max_len=90) for s in range(3)]
    plain = Trainer(TrainConfig(**job), cfg, torch.device(DEV))
    ref = [float(plain.train_step(plain.next_batch(b))["loss_per_sample"]) for b in batches]
    ref_norm = float(plain.train_step(plain.next_batch(batches[0]))["grad_norm"])
    monkeypatch.setenv("TN_FORCE_FSDP", "1")
    monkeypatch.setenv("TN_DP_FORCE_COLLECTIVES", "1")       # (a lone rank would otherwise skip the identity collectives)
    tr = Trainer(TrainConfig(**job, training_dp_engine="flat"), cfg, torch.device(DEV, 0), dp_mesh=build_dp_mesh("cuda", 1))
    eng = tr.dp_engine
    assert eng is not None and eng.world == 1 and not eng.identity and len(eng.buckets) == 2 + 3
    assert all(p.dtype == torch.bfloat16 and not hasattr(p, "_local_tensor") for p in tr.model.parameters())
    got = [float(tr.train_step(tr.next_batch(b))["loss_per_sample"]) for b in batches]
    norm = float(tr.train_step(tr.next_batch(batches[0]))["grad_norm"])
    assert eng.buckets[0].shard.grad.dtype == getattr(torch, reduce)
    assert eng.sunk == 2 * 7              # (registered; at these widths the products are too small for the own GEMM)
    assert all(p.grad is None for p in tr.model.parameters())
    tol = 1e-5 if reduce == "float32" else 2e-3            # same init, same kernels: only the gradient dtype differs
    for a, b in zip(got, ref):
        assert abs(a - b) / abs(b) < tol, (got, ref)
    # (the fourth step's gradient norm sits behind three optimizer steps taken from bf16 vs fp32 gradients: the two
    #  trajectories agree to 1e-5 in the loss and to a few 1e-4 in the norm — 0.5e-4 .. 2.4e-4 depending on which attention
    #  forward kernel rounds the row sums, both values reproducible to the last digit, profiles/r06b_*)
    assert abs(norm - ref_norm) / ref_norm < max(tol, 5e-4), (norm, ref_norm)


def test_flat_engine_through_the_reference_hooks_on_device(rccl_single_rank, monkeypatch):
    """The reference trainer's sequence (boundary.json: parallelize_fn on a meta model -> to_empty -> post_init ->
    .to(float32) -> build_optimizers_fn; per step zero_grad / backward / clip_grad_norm_ / step) with the real kernels:
    `parallelize_fn` marks, `build_optimizers_fn` builds the flat engine on bf16 views + FusedAdamW whose masters are the
    trainer's float32 numbers; three steps follow the plain trainer's losses and the reference's own clip sees no gradient."""
    import touchnet_amd.specs as specs
    from touchnet_amd.bin.train import TrainConfig, Trainer, _MeshView
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig
    from touchnet_amd.utils.distributed import ParallelDims, build_dp_mesh
    from touchnet_amd.utils.train_spec import get_train_spec
    from touchnet_amd.utils.zero_dp import FlatEngineOptimizer
    cfg = DecoderConfig.from_dict(dict(CFG, model_type="llama"))
    kw = dict(training_model_name="llama_mi355", training_enable_fused_ce=True, lr_scheduler_warmup_steps=0,
              lr_scheduler_lr=1e-3)
    batches = [text_batch(1024, 4, 512, seed=s, max_len=90) for s in range(3)]
    plain = Trainer(TrainConfig(**kw), cfg, torch.device(DEV))
    ref = [float(plain.train_step(plain.next_batch(b))["loss_per_sample"]) for b in batches]
    monkeypatch.setenv("TN_DP_FORCE_COLLECTIVES", "1")
    job = TrainConfig(**kw, training_dp_engine="flat")
    spec = get_train_spec("llama_mi355")
    mesh = build_dp_mesh("cuda", 1)
    dims = ParallelDims(dp_replicate=1, dp_shard=2, cp=1, tp=1, pp=1, world_size=2, enable_loss_parallel=False)  # (takes the dp branch)
    dev = torch.device(DEV, 0)
    torch.manual_seed(job.training_seed)
    with torch.device("meta"):
        model = spec.model_cls(cfg)
    model = spec.parallelize_fn(model, _MeshView({"dp_shard_cp": mesh}), dims, job)
    assert all(p.is_meta for p in model.parameters())
    model.to_empty(device=dev)
    with torch.no_grad():
        model.post_init()
        spec.additional_post_init_fn(model, dev)
    model.train()
    model = model.to(torch.float32)
    f32 = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt = spec.build_optimizers_fn([model], job)
    lrs = spec.build_lr_schedulers_fn(opt, job)
    assert isinstance(opt, FlatEngineOptimizer) and not opt.engine.identity
    assert all(p.dtype == torch.bfloat16 for p in model.parameters())
    # the masters are the float32 numbers, not the rounded ones
    b0 = opt.engine.buckets[0]
    m0 = opt.inner.state[0]["master"]
    p0, o0 = b0.params[0], b0.offsets[0]
    name0 = next(n for n, p in model.named_parameters() if p is p0)
    assert torch.equal(m0[o0:o0 + p0.numel()], f32[name0].reshape(-1))
    got = []
    for b in batches:
        data = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        ns = torch.tensor([float(data.pop("num_sentence"))], device=dev)
        opt.zero_grad()
        pred = model(**data, num_sentence=ns, ce_chunk_tokens=job.training_ce_chunk_tokens)
        pred.loss.backward()
        assert all(p.grad is None for p in model.parameters())
        norm = torch.nn.utils.clip_grad_norm_(list(model.parameters()), job.training_max_norm)
        assert float(norm) == 0.0
        opt.step()
        lrs.step()
        got.append(float(pred.loss))
        assert float(opt.last_grad_norm) > 0
    # (the plain trainer starts from bf16-rounded masters, this one from the float32 draw: same numbers to bf16 rounding)
    for a, b in zip(got, ref):
        assert abs(a - b) / abs(b) < 2e-3, (got, ref)


def _lp_worker(rank, world, port, ret):
    """one of two processes on the SAME GPU (gloo carries the device tensors of the tiny statistics exchanges)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import touchnet_amd.functional as F
        g = torch.Generator().manual_seed(11)
        n, H, V = 700, 256, 4096
        h = (torch.randn(1, n, H, generator=g) * 0.5).bfloat16().to(DEV)
        w = (torch.randn(V, H, generator=g) * 0.2).bfloat16().to(DEV)
        labels = torch.randint(0, V, (1, n), generator=g)
        labels[0, ::3] = -100
        labels[0, 5] = V - 1
        labels[0, 7] = V // 2                                        # first row of the second shard
        sl = torch.randint(1, 9, (1, n), generator=g)
        labels, sl = labels.to(DEV), sl.to(DEV)
        # single-process truth on the full head
        hf = h.clone().requires_grad_()
        wf = w.clone().requires_grad_()
        loss, stats = F.fused_linear_cross_entropy(hf, wf, labels, sl, 37, -100, 256)
        loss.backward()
        # this rank's vocabulary shard, three chunks, with and without the compact row selection
        out = {}
        for compact in (False, 512):
            hl = h.clone().requires_grad_()
            wl = w[rank * V // world:(rank + 1) * V // world].clone().requires_grad_()
            l2, s2 = F.fused_linear_cross_entropy(hl, wl, labels, sl, 37, -100, 256, compact,
                                                  tp=(dist.group.WORLD, rank, world))
            l2.backward()
            dh = hl.grad.float().clone()
            dist.all_reduce(dh)                                      # (in the model: the sequence gather's reduce-scatter)
            out[compact] = (abs(float(l2) - float(loss)) / abs(float(loss)),
                            float((s2 - stats).abs().max()),
                            float((dh - hf.grad.float()).abs().max()) / float(hf.grad.float().abs().max()),
                            float((wl.grad.float() - wf.grad.float()[rank * V // world:(rank + 1) * V // world]).abs().max())
                            / float(wf.grad.float().abs().max()))
        ret[rank] = ("ok", out)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_vocabulary_parallel_fused_lm_head_ce_on_the_hip_kernels_two_processes_one_gpu():
    """Loss parallel through the product path (functional._FusedLinearCE with tp=): local logits + row statistics from the
    unchanged CE kernels, max / sum-exp / target / argmax combined over the group, backward with the global log-sum-exp.
    Two processes share the one GPU (their exchange goes over gloo); loss, statistics (incl. accuracy), d(hidden) summed
    over the shards and each shard's d(weight) equal the single-process fused lm_head + CE."""
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_lp_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
        results = dict(ret)
    for r in range(2):
        assert r in results and results[r][0] == "ok", results.get(r)
        for compact, (e_loss, e_stats, e_dh, e_dw) in results[r][1].items():
            assert e_loss < 1e-5 and e_stats < 1e-4, (r, compact, results[r][1])
            assert e_dh < 2e-2 and e_dw < 2e-2, (r, compact, results[r][1])      # bf16 GEMMs of different shapes


def _flat2_worker(rank, world, port, ref_losses, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import touchnet_amd.specs  # noqa: F401
        from touchnet_amd.bin.train import TrainConfig, Trainer
        from touchnet_amd.data.synthetic import text_batch
        from touchnet_amd.models.llama import DecoderConfig
        torch.cuda.set_device(0)

        class mesh:                  # (a cuda DeviceMesh would create an RCCL group: two ranks cannot share one GPU there)
            get_group = staticmethod(lambda: dist.group.WORLD)
            size = staticmethod(lambda: world)
        cfg = DecoderConfig.from_dict(dict(CFG, model_type="llama"))
        job = TrainConfig(training_model_name="llama_mi355", training_enable_fused_ce=True, lr_scheduler_warmup_steps=0,
                          lr_scheduler_lr=1e-3, training_dp_engine="flat", training_max_norm=1e9)
        tr = Trainer(job, cfg, torch.device(DEV, 0), dp_mesh=mesh)
        eng = tr.dp_engine
        from touchnet_amd import _C
        # (collectives beside the compute: one workgroup per tile, chosen by a library call, not through the environment)
        assert eng is not None and eng.world == 2 and not eng.identity and _C.lib().tn_gemm_get_persistent() == 0
        assert "TN_GEMM_PERSIST" not in os.environ
        losses = []
        for s in range(3):
            b = text_batch(1024, 2, 512, seed=10 * s + rank, max_len=90)          # each rank its own rows
            losses.append(float(tr.train_step(tr.next_batch(b))["loss_per_sample"]))
        torch.cuda.synchronize()
        # the replicas stay identical (every rank gathered the same updated slices)
        flat = torch.cat([b.flat_p.float().reshape(-1)[:4096] for b in eng.buckets])
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        ret[rank] = ("ok", losses, bool(torch.equal(both[0], both[1])), len(eng._pool))
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_flat_engine_two_ranks_on_one_gpu_trains_like_one_process_on_the_joined_batch():
    """World size 2 on the device (two processes share the GPU, gloo moves the device buffers): staging pool, side-stream
    reduce-scatter (AVG) and in-place all-gather with REAL peers around the HIP kernels.  The per-rank losses equal those of
    one unsharded process that sees both ranks' rows (B = 4 instead of 2 x 2): the same global num_sentence normalisation, the
    dp-averaged gradient equals the joined batch's gradient / 2 — AdamW is invariant to that factor, the clip is not reached."""
    import socket

    import torch.multiprocessing as mp
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig
    cfg = DecoderConfig.from_dict(dict(CFG, model_type="llama"))
    job = TrainConfig(training_model_name="llama_mi355", training_enable_fused_ce=True, lr_scheduler_warmup_steps=0,
                      lr_scheduler_lr=1e-3, training_max_norm=1e9)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_flat2_worker, args=(r, 2, port, None, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(900)
        results = dict(ret)
    for r in range(2):
        assert r in results and results[r][0] == "ok", results.get(r)
        assert results[r][2], "replicas diverged"
        assert results[r][3] <= 3
    # one process on the joined rows: its loss (sum over all sentences / global count) = sum of the ranks' loss parts
    plain = Trainer(job, cfg, torch.device(DEV))
    for s in range(3):
        parts = [text_batch(1024, 2, 512, seed=10 * s + r, max_len=90) for r in range(2)]
        joined = {k: (torch.cat([p[k] for p in parts]) if isinstance(parts[0][k], torch.Tensor)
                      else sum(p[k] for p in parts) if isinstance(parts[0][k], int) else parts[0][k]) for k in parts[0]}
        ref = float(plain.train_step(plain.next_batch(joined))["loss_per_sample"])
        got = results[0][1][s] + results[1][1][s]
        assert abs(got - ref) / abs(ref) < 3e-3, (s, got, ref)


def _cp2_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import touchnet_amd.specs  # noqa: F401
        from touchnet_amd.bin.train import TrainConfig, Trainer
        from touchnet_amd.data.synthetic import text_batch
        from touchnet_amd.models.llama import DecoderConfig
        torch.cuda.set_device(0)

        class mesh:                  # (see _flat2_worker)
            get_group = staticmethod(lambda: dist.group.WORLD)
            size = staticmethod(lambda: world)
        solo = [dist.new_group([r]) for r in range(world)][rank]      # dp = 1: `num_sentence` is summed over THIS group

        class dp_mesh:
            get_group = staticmethod(lambda: solo)
            size = staticmethod(lambda: 1)
        cfg = DecoderConfig.from_dict(dict(CFG, model_type="llama"))
        job = TrainConfig(training_model_name="llama_mi355", training_enable_fused_ce=True, lr_scheduler_warmup_steps=0,
                          lr_scheduler_lr=1e-3, training_dp_engine="flat", training_max_norm=1e9)
        tr = Trainer(job, cfg, torch.device(DEV, 0), dp_mesh=dp_mesh, cp_mesh=mesh, fsdp_mesh=mesh)
        losses, halo = [], 0
        for s in range(3):
            b = text_batch(1024, 1, 2048, seed=50 + s, min_len=200, max_len=700)      # the SAME rows on both cp ranks
            data = tr.next_batch(b)
            assert data["input_ids"].shape == (1, 1024) and data["attention_mask"].shape == (1, 2048)
            losses.append(float(tr.train_step(data)["loss_per_sample"]))
            halo = max(halo, tr.cp.halo_bytes)
        torch.cuda.synchronize()
        ret[rank] = ("ok", losses, halo)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_context_parallel_two_ranks_on_one_gpu_trains_like_one_process():
    """cp = 2 on the device with a REAL peer (two processes share the GPU; gloo carries the K/V chunks and the returned
    dK/dV): zig-zag sequence shards, halo exchange issued before the query path, own-chunk attention, wait, received-chunk
    attention, fp32 LSE merge, single segment backward, halo return, gradients reduced by the flat engine over the cp group.
    The loss parts of the two ranks add up to the loss of ONE process on the whole sequence, three optimizer steps in a row."""
    import socket

    import torch.multiprocessing as mp
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_cp2_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(900)
        results = dict(ret)
    for r in range(2):
        assert r in results and results[r][0] == "ok", results.get(r)
    assert results[0][2] > 0 or results[1][2] > 0                      # documents cross the chunk boundaries: K/V really moved
    cfg = DecoderConfig.from_dict(dict(CFG, model_type="llama"))
    job = TrainConfig(training_model_name="llama_mi355", training_enable_fused_ce=True, lr_scheduler_warmup_steps=0,
                      lr_scheduler_lr=1e-3, training_max_norm=1e9)
    plain = Trainer(job, cfg, torch.device(DEV))
    for s in range(3):
        b = text_batch(1024, 1, 2048, seed=50 + s, min_len=200, max_len=700)
        ref = float(plain.train_step(plain.next_batch(b))["loss_per_sample"])
        got = results[0][1][s] + results[1][1][s]
        assert abs(got - ref) / abs(ref) < 3e-3, (s, got, ref)


def _tp2_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from touchnet_amd.models.tensor_parallel import apply_tp, reduce_sequence_partial_grads
        torch.cuda.set_device(0)

        class mesh:
            get_group = staticmethod(lambda: dist.group.WORLD)
            size = staticmethod(lambda: world)
            get_local_rank = staticmethod(lambda: rank)
        batch = _packed_batch(2, 1024, 1024, seed=8)
        full = _model(seed=9)
        loss0, g0 = _loss_and_grads(full, batch)
        out = {}
        for name, kw in (("plain", {}), ("sp", dict(sequence_parallel=True)),
                         ("sp+lp", dict(sequence_parallel=True, loss_parallel=True))):
            part = apply_tp(_model(seed=9), mesh, **kw)
            loss1, _ = _loss_and_grads(part, batch)
            reduce_sequence_partial_grads(part)
            worst = 0.0
            for n, p in part.named_parameters():
                want = g0[n]
                if p.shape != want.shape:
                    want = want.chunk(world, dim=1 if ("o_proj" in n or "down_proj" in n) else 0)[rank]
                worst = max(worst, _rel(p.grad, want))
            out[name] = (abs(loss1 - loss0) / abs(loss0), worst, len(part._tn_tp["sharded_names"]))
        ret[rank] = ("ok", out)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_tensor_parallel_two_ranks_on_one_gpu_through_the_hip_kernels():
    """tp = 2 with a REAL peer on the device (two processes share the GPU, gloo carries the all-reduces / sequence gathers /
    reduce-scatters): replicated residual stream, sequence parallel, and sequence + loss parallel (vocabulary-sharded head in
    the fused lm_head + CE) — loss and every gradient (shards against the matching slices) equal the unsharded model run by
    the same process on the same kernels."""
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_tp2_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(900)
        results = dict(ret)
    for r in range(2):
        assert r in results and results[r][0] == "ok", results.get(r)
        for name, (e_loss, e_grad, n_sharded) in results[r][1].items():
            assert e_loss < 2e-3 and e_grad < 5e-2, (r, name, results[r][1])
            assert n_sharded == 2 * 10 + (1 if name == "sp+lp" else 0)


def test_flat_engine_weight_gradients_written_by_the_gemm_into_fp32_staging(rccl_single_rank, monkeypatch):
    """At widths the hand-written GEMM takes (2560-wide layers: 100 tiles, split-K in 2), the blocks' weight-gradient GEMMs
    write the engine's fp32 reduce-scatter input themselves (`functional.GRAD_SINKS`, tn_gemm_bf16_wgrad_f32) — no bf16
    gradient tensor, no cast-copy; the steps equal the plain trainer's."""
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig
    from touchnet_amd.utils.distributed import build_dp_mesh
    wide = dict(CFG, model_type="llama", hidden_size=2560, intermediate_size=2560, num_attention_heads=20,
                num_key_value_heads=20, head_dim=128, num_hidden_layers=2)
    cfg = DecoderConfig.from_dict(wide)
    job = dict(training_model_name="llama_mi355", training_enable_fused_ce=True, lr_scheduler_warmup_steps=0,
               lr_scheduler_lr=1e-3)
    batches = [text_batch(1024, 4, 512, seed=s, max_len=90) for s in range(3)]
    plain = Trainer(TrainConfig(**job), cfg, torch.device(DEV))
    ref = [float(plain.train_step(plain.next_batch(b))["loss_per_sample"]) for b in batches]
    del plain
    monkeypatch.setenv("TN_FORCE_FSDP", "1")
    monkeypatch.setenv("TN_DP_FORCE_COLLECTIVES", "1")
    tr = Trainer(TrainConfig(**job, training_dp_engine="flat"), cfg, torch.device(DEV, 0), dp_mesh=build_dp_mesh("cuda", 1))
    got = [float(tr.train_step(tr.next_batch(b))["loss_per_sample"]) for b in batches]
    eng = tr.dp_engine
    linear = sum(p.numel() for blk in tr.model.model.layers for n, p in blk.named_parameters() if p.dim() == 2)
    total = sum(p.numel() for p in tr.model.parameters())
    assert eng.sunk == 14 and eng.staged_bytes == (total - linear) * 6          # only norms / embeddings / head were cast-copied
    # (the engine's weight gradients are the UNROUNDED fp32 accumulators, the plain trainer's are rounded to bf16 first: the
    #  first step is identical, AdamW then amplifies the last-bit differences — 3e-5 at step 2, 5e-4 at step 3 observed)
    assert got[0] == pytest.approx(ref[0], rel=1e-6)
    for a, b in zip(got, ref):
        assert abs(a - b) / abs(b) < 3e-3, (got, ref)


@pytest.mark.parametrize("engine,bias", [("plain", False), ("flat", False), ("plain", True), ("flat", True)])
def test_weight_gradients_on_a_side_stream_train_identically(rccl_single_rank, engine, bias, monkeypatch):
    """functional.enable_wgrad_stream(): the weight-gradient GEMMs (own kernel, 2560-wide layers) run on a second stream
    beside the input-gradient chain, inputs kept alive by record_stream, consumers (optimizer, the flat engine's
    reduce-scatter) waiting for that stream — the same numbers as the single-stream run, step for step.  `bias` (Qwen2's
    q / k / v biases): the bias gradient is allocated on the main stream, FILLED by the weight-gradient launch
    on the side stream and read by autograd on the main stream — `_beside(written=...)` makes the main stream wait; the
    grouped MLP launch returns a LIST of gradients, which gets the same treatment as a single tensor."""
    import touchnet_amd.functional as F
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.llama import DecoderConfig
    from touchnet_amd.utils.distributed import build_dp_mesh
    wide = dict(CFG, model_type="llama", hidden_size=2560, intermediate_size=2560, num_attention_heads=20,
                num_key_value_heads=20, head_dim=128, num_hidden_layers=2, attention_bias=bias)
    cfg = DecoderConfig.from_dict(wide)
    job = dict(training_model_name="llama_mi355", training_enable_fused_ce=True, lr_scheduler_warmup_steps=0,
               lr_scheduler_lr=1e-3)
    batches = [text_batch(1024, 4, 512, seed=s, max_len=90) for s in range(4)]
    if engine == "flat":
        monkeypatch.setenv("TN_FORCE_FSDP", "1")
        monkeypatch.setenv("TN_DP_FORCE_COLLECTIVES", "1")
        job["training_dp_engine"] = "flat"
    make = lambda: Trainer(TrainConfig(**job), cfg, torch.device(DEV, 0),
                           dp_mesh=build_dp_mesh("cuda", 1) if engine == "flat" else None)

    def run():
        tr = make()
        out = [float(tr.train_step(tr.next_batch(b))["loss_per_sample"]) for b in batches]
        norm = float(tr.train_step(tr.next_batch(batches[0]))["grad_norm"])
        return out, norm
    ref = run()
    monkeypatch.setenv("TN_GEMM_PERSIST", "0")
    F.enable_wgrad_stream(True)
    try:
        got = run()
    finally:
        F.enable_wgrad_stream(False)
    assert got == ref, (got, ref)
