"""N > 1 path on CPU: 2 ranks over gloo, FSDP2 (`fully_shard`) wrapping + the step driver's collectives.

What is under test is the HOST logic the 8-GPU run relies on (mesh, FSDP grouping incl. tied embeddings,
global `num_sentence` all-reduce, loss/grad scaling convention, sharded optimizer plumbing); the per-op
arithmetic is the oracle's, injected explicitly (the product has no CPU path)."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

TINY = dict(vocab_size=16, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=8,
            num_key_value_heads=4, head_dim=8, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=True,
            initializer_range=0.2)


class TorchAdamW:
    """torch.optim.AdamW behind the FusedAdamW interface (CPU stand-in used only by this test)."""

    def __init__(self, params, lr=1e-2, max_norm=1.0):
        self.params = [p for p in params]
        self.opt = torch.optim.AdamW(self.params, lr=lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
        self.max_norm = max_norm

    def zero_grad(self):
        self.opt.zero_grad(set_to_none=True)

    def step(self, lr=None):
        norm = torch.nn.utils.clip_grad_norm_(self.params, self.max_norm)
        self.opt.step()
        return norm.full_tensor() if hasattr(norm, "full_tensor") else norm


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batches(world):
    from touchnet_amd.data.synthetic import text_batch
    return [text_batch(16, 2, 32, seed=100 + r, max_len=9) for r in range(world)]


def _reference(world):
    """Single-process ground truth: global loss = sum_r local(per-rank) parts / global num_sentence; FSDP's
    reduce-scatter AVERAGES gradients over dp, i.e. grad = d(global loss)/dW / world (train.py:456 convention)."""
    import oracle.ops as oops
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    torch.manual_seed(7)
    model = PackedCausalLM(DecoderConfig.from_dict(TINY))
    model.post_init()
    batches = _batches(world)
    total_ns = sum(b["num_sentence"] for b in batches)
    losses = []
    with use_ops(oops):
        for b in batches:
            out = model(input_ids=b["input_ids"], position_ids=b["position_ids"], attention_mask=b["attention_mask"],
                        labels=b["labels"], sentence_lens=b["sentence_lens"], num_sentence=total_ns)
            losses.append(out.loss)
        (sum(losses) / world).backward()
    return ({k: v.detach().clone() for k, v in model.state_dict().items()},
            {n: p.grad.clone() for n, p in model.named_parameters()}, [float(l) for l in losses], total_ns)


def _worker(rank, world, port, ref_state, ref_grads, ref_losses, total_ns, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle.ops as oops
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.llama import DecoderConfig
    from touchnet_amd.utils.distributed import build_dp_mesh, init_distributed
    try:
        r, _, w = init_distributed("cpu")
        assert (r, w) == (rank, world)
        mesh = build_dp_mesh("cpu", world)
        job = TrainConfig(training_model_name="llama_mi355", training_enable_fused_ce=True,
                          training_mixed_precision_param="float32", training_fsdp_reshard_after_forward="never")
        with use_ops(oops):
            tr = Trainer(job, DecoderConfig.from_dict(TINY), torch.device("cpu"), dp_mesh=mesh,
                         optimizer_factory=lambda ps: TorchAdamW(ps))
            # tied embedding survives meta-init + fully_shard + to_empty
            lm = tr.model
            assert lm.lm_head.weight is lm.model.embed_tokens.weight
            # load the reference weights into the shards
            with torch.no_grad():
                for name, p in tr.model.named_parameters():
                    full = ref_state[name]
                    local = p._local_tensor if hasattr(p, "_local_tensor") else p
                    local.copy_(full.chunk(world, dim=0)[rank] if local.shape != full.shape else full)
            batch = _batches(world)[rank]
            data = tr.next_batch(batch)
            assert float(data["num_sentence"]) == float(total_ns)          # global SUM over dp
            # dev loop (train.py:553-621): no_grad forward, metrics reduced over dp: the loss parts ADD UP
            m = tr.dev([batch, batch])
            assert m["batches"] == 2 and tr.model.training
            assert float(m["global_avg_loss_per_sample"]) == pytest.approx(sum(ref_losses), rel=1e-5)
            assert float(m["global_max_loss_per_token"]) >= float(m["global_avg_loss_per_token"]) > 0
            assert 0.0 <= float(m["global_min_acc"]) <= float(m["global_avg_acc"]) <= 1.0
            assert all(p.grad is None for p in tr.model.parameters())
            tr.optimizer.zero_grad()
            loss, per_token, acc = tr.forward_loss(data)
            assert float(loss) == pytest.approx(ref_losses[rank], rel=1e-5, abs=1e-6)
            loss.backward()
            worst = 0.0
            for name, p in tr.model.named_parameters():
                g = p.grad.full_tensor() if hasattr(p.grad, "full_tensor") else p.grad
                worst = max(worst, float((g - ref_grads[name]).abs().max()))
            assert worst < 2e-5, worst
            stats = tr.train_step(data)                                    # full step incl. optimizer on shards
            assert torch.isfinite(stats["grad_norm"]).all()
        ret[rank] = ("ok", float(loss), worst)
    except Exception as e:  # surface the failure in the parent
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_fsdp2_step_two_ranks_gloo(world):
    ref_state, ref_grads, ref_losses, total_ns = _reference(world)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ref_state, ref_grads, ref_losses, total_ns, ret), nprocs=world,
                 join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]


# --------------------------------------------------------------------------------------------------------
# context parallel: 2 ranks share one packed batch along the sequence dimension
# --------------------------------------------------------------------------------------------------------
CP_CFG = dict(TINY, num_hidden_layers=1)


def _cp_reference(B=1, T=512):
    import oracle.ops as oops
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    torch.manual_seed(11)
    model = PackedCausalLM(DecoderConfig.from_dict(CP_CFG))
    model.post_init()
    batch = text_batch(16, B, T, seed=5, max_len=90)
    with use_ops(oops):
        out = model(input_ids=batch["input_ids"], position_ids=batch["position_ids"],
                    attention_mask=batch["attention_mask"], labels=batch["labels"],
                    sentence_lens=batch["sentence_lens"], num_sentence=batch["num_sentence"])
        out.loss.backward()
    return ({k: v.detach().clone() for k, v in model.state_dict().items()},
            {n: p.grad.clone() for n, p in model.named_parameters()}, float(out.loss), batch)


def _cp_worker(rank, world, port, ref_state, ref_grads, ref_loss, batch, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle.ops as oops
    import touchnet_amd.specs  # noqa: F401
    from torch.distributed.device_mesh import init_device_mesh
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.llama import DecoderConfig
    from touchnet_amd.utils.distributed import init_distributed
    try:
        init_distributed("cpu")
        mesh = init_device_mesh("cpu", (1, world), mesh_dim_names=("dp", "cp"))
        flat = mesh["dp", "cp"]._flatten("dp_cp")
        job = TrainConfig(training_model_name="llama_mi355", training_enable_fused_ce=True,
                          training_mixed_precision_param="float32")
        with use_ops(oops):
            tr = Trainer(job, DecoderConfig.from_dict(CP_CFG), torch.device("cpu"), dp_mesh=mesh["dp"],
                         cp_mesh=mesh["cp"], fsdp_mesh=flat, optimizer_factory=lambda ps: TorchAdamW(ps))
            with torch.no_grad():
                for name, p in tr.model.named_parameters():
                    full = ref_state[name]
                    local = p._local_tensor if hasattr(p, "_local_tensor") else p
                    local.copy_(full.chunk(world, dim=0)[rank] if local.shape != full.shape else full)
            data = tr.next_batch(batch)
            T = batch["input_ids"].shape[1]
            assert data["input_ids"].shape[1] == T // world and data["attention_mask"].shape[1] == T
            # head/tail load balancing: rank r holds chunks r and 2cp-1-r
            Tc = T // (2 * world)
            exp = torch.cat([batch["input_ids"][:, rank * Tc:(rank + 1) * Tc],
                             batch["input_ids"][:, (2 * world - 1 - rank) * Tc:(2 * world - rank) * Tc]], dim=1)
            assert torch.equal(data["input_ids"], exp)
            tr.optimizer.zero_grad()
            loss, _, _ = tr.forward_loss(data)
            total = loss.detach().clone()
            dist.all_reduce(total)                                       # loss parts of the cp ranks add up
            assert float(total) == pytest.approx(ref_loss, rel=1e-5)
            loss.backward()
            worst = 0.0
            for name, p in tr.model.named_parameters():
                g = p.grad.full_tensor() if hasattr(p.grad, "full_tensor") else p.grad
                # FSDP AVERAGES over the dp_cp mesh: (1/cp) * sum of the shard gradients = ref / cp
                worst = max(worst, float((g * world - ref_grads[name]).abs().max()))
            assert worst < 3e-5, worst
        ret[rank] = ("ok", float(total), worst)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


@pytest.mark.parametrize("world,B,T", [(2, 1, 512), (4, 1, 1024), (2, 2, 512)])
def test_context_parallel_full_step_gloo(world, B, T):
    """A full forward + loss + backward under context parallelism (halo exchange issued before the query path, finished
    in front of the attention; dK/dV returned under the query-path backward) == the single-process run: the loss parts
    of the cp ranks add up, FSDP-averaged gradients x cp == the reference gradients.  B = 1: chunks are received
    straight into the global K/V buffers; B = 2: through staging buffers."""
    ref_state, ref_grads, ref_loss, batch = _cp_reference(B, T)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_cp_worker, args=(world, _free_port(), ref_state, ref_grads, ref_loss, batch, ret), nprocs=world,
                 join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]


@pytest.mark.parametrize("cp", [2, 4])
def test_halo_need_matrix_against_elementwise_mask(cp):
    """halo_need() vs a brute-force evaluation of the document mask: a rank needs a chunk exactly when some of its
    query rows may attend some key of that chunk (plus its own chunks)."""
    from touchnet_amd.utils.context_parallel import halo_need
    rng = np.random.RandomState(cp)
    for trial in range(20):
        B, Tc = 2, 8
        T = 2 * cp * Tc
        ids = np.zeros((B, T), dtype=np.int64)
        for b in range(B):
            t, d = 0, 1
            while t < T:
                n = int(rng.randint(1, 3 * Tc))
                ids[b, t:t + n] = d if rng.rand() > 0.15 else 0
                t += n
                d += 1
        need = halo_need(ids, cp)
        q = np.arange(T)
        allow = (ids[:, :, None] == ids[:, None, :]) & (ids[:, :, None] > 0) & (q[None, None, :] <= q[None, :, None])
        C = 2 * cp
        for r in range(cp):
            mine = (r, C - 1 - r)
            for c in range(C):
                rows = np.concatenate([np.arange(lc * Tc, (lc + 1) * Tc) for lc in mine])
                brute = bool(allow[:, rows][:, :, c * Tc:(c + 1) * Tc].any()) or c in mine
                assert bool(need[r, c]) == brute, (trial, r, c)


def _halo_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from touchnet_amd.utils.context_parallel import ContextParallel
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        T, B, Hd = 2 * world * 128, 2, 3
        cpar = ContextParallel(dist.group.WORLD, T)
        g = torch.Generator().manual_seed(5)
        full = torch.randn(B, T, Hd, generator=g)
        w = torch.randn(B, T, Hd, generator=g)                       # fixed weights of a dummy loss
        # documents aligned to the chunks: chunk c holds document c+1 only -> nobody needs anybody else's chunks;
        # second layout: one document spans everything -> every earlier chunk is needed
        outs = {}
        for name, ids in (("disjoint", torch.arange(2 * world).repeat_interleave(128)[None].repeat(B, 1) + 1),
                          ("one_doc", torch.ones(B, T, dtype=torch.int64))):
            res = {}
            for mode in ("allgather", "halo"):
                cpar.need, cpar.halo_bytes = None, 0
                if mode == "halo":
                    cpar.set_documents(ids)
                x = cpar.shard(full).clone().requires_grad_()
                y = cpar.gather_seq(x)
                # only what the document mask lets this rank's rows see may matter: weight the visible chunks
                vis = torch.zeros(T)
                need = cpar.need if mode == "halo" else None
                from touchnet_amd.utils.context_parallel import halo_need
                nd = halo_need(ids.numpy(), world)
                for c in range(2 * world):
                    if nd[rank, c]:
                        vis[c * cpar.Tc:(c + 1) * cpar.Tc] = 1.0
                (y * w * vis[None, :, None]).sum().backward()
                res[mode] = (y.detach() * vis[None, :, None], x.grad.clone(), cpar.halo_bytes)
            assert torch.equal(res["halo"][0], res["allgather"][0]), name
            assert torch.allclose(res["halo"][1], res["allgather"][1], atol=1e-6), name
            outs[name] = res["halo"][2]
        ret[rank] = ("ok", outs)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_halo_exchange_equals_allgather_and_trims(world):
    """gloo, 2 and 4 ranks: gather_seq through the halo exchange == through all-gather/reduce-scatter on everything a
    rank's rows can see (values and gradients); chunk-aligned documents move NOTHING, one long document moves the
    causal prefix only (byte counts derived from halo_need: forward receives + backward gradient returns)."""
    from touchnet_amd.utils.context_parallel import halo_need
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_halo_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]
    chunk_bytes = 2 * 128 * 3 * 4
    need = halo_need(np.ones((2, 2 * world * 128), dtype=np.int64), world)
    C = 2 * world
    for r in range(world):
        assert results[r][1]["disjoint"] == 0
        mine = (r, C - 1 - r)
        fwd = sum(bool(need[r, c]) for c in range(C) if c not in mine)            # chunks received in the forward
        bwd = sum(bool(need[p, c]) for p in range(world) if p != r for c in mine)  # gradient pieces coming back
        assert results[r][1]["one_doc"] == (fwd + bwd) * chunk_bytes, (r, results[r][1], fwd, bwd)


@pytest.mark.parametrize("cp", [2, 4])
def test_device_side_halo_table_equals_exact_one_on_packer_ids(cp):
    """halo_need_ranges (torch ops on chunk-level id ranges, runs on the device without a host round trip) == halo_need
    (exact, host) for ids that do not decrease along a row — what every packer emits —, and is a SUPERSET of it for
    arbitrary ids (never drops a chunk a rank needs)."""
    from touchnet_amd.utils.context_parallel import halo_need, halo_need_ranges
    rng = np.random.RandomState(cp)
    T = 2 * cp * 128
    for trial in range(20):
        B = int(rng.randint(1, 4))
        ids = np.zeros((B, T), dtype=np.int64)
        for b in range(B):
            t, d = 0, 1
            end = T - int(rng.randint(0, 200))
            while t < end:
                n = int(rng.randint(1, [40, 300, 2000][trial % 3]))
                ids[b, t:min(t + n, end)] = d
                t += n
                d += 1
        exact = halo_need(ids, cp)
        got = halo_need_ranges(torch.from_numpy(ids), cp).numpy()
        assert np.array_equal(got, exact), trial
        shuffled = ids.copy()
        perm = rng.permutation(int(ids.max()) + 1)
        perm = np.concatenate([[0], 1 + rng.permutation(int(ids.max()))])       # relabel documents, keep 0 = pad
        shuffled = perm[ids]
        got = halo_need_ranges(torch.from_numpy(shuffled), cp).numpy()
        assert not (halo_need(shuffled, cp) & ~got).any(), trial


def _order_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle.ops as oops
        from touchnet_amd.models.backend import use_ops
        from touchnet_amd.models.llama import DecoderConfig
        from touchnet_amd.models.llama.modeling_llama import Attention, RotaryEmbedding
        from touchnet_amd.utils import context_parallel as CP
        cfg = DecoderConfig(vocab_size=16, hidden_size=64, intermediate_size=128, num_hidden_layers=1,
                            num_attention_heads=4, num_key_value_heads=2, head_dim=16)
        T = 512
        cp = CP.ContextParallel(dist.group.WORLD, T)
        doc = torch.ones(1, T, dtype=torch.int64)
        cp.set_documents(doc)
        order = []
        for name, cls in (("issue_return", CP._HaloReturn), ("finish_return", CP._HaloStart)):
            orig = cls.backward

            def wrap(ctx, *a, _o=orig, _n=name):
                order.append(_n)
                return _o(ctx, *a)
            cls.backward = staticmethod(wrap)
        torch.manual_seed(0)
        attn = Attention(cfg)
        attn.q_proj.weight.register_hook(lambda g: order.append("q_proj_backward"))
        attn.k_proj.weight.register_hook(lambda g: order.append("k_proj_backward"))
        with use_ops(oops):
            x = torch.randn(1, T // world, 64, requires_grad=True)
            cos, sin = RotaryEmbedding(cfg)(cp.shard(torch.arange(T)[None]), torch.float32)
            mask = oops.build_packed_mask(doc)
            mask.cp = cp
            attn(x, cos, sin, mask).sum().backward()
        ret[rank] = ("ok", order)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def _split_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle.ops as oops
        from touchnet_amd.models.backend import use_ops
        from touchnet_amd.utils import context_parallel as CP
        T, B, Nh, Nkv, D = 2 * world * 128, 2, 4, 2, 16
        cp = CP.ContextParallel(dist.group.WORLD, T)
        g = torch.Generator().manual_seed(11)
        # two documents per row, the second one starts in the middle of a chunk; a pad tail
        doc = torch.ones(B, T, dtype=torch.int64)
        doc[:, 200:] = 2
        doc[1, T - 40:] = 0
        cp.set_documents(doc)
        q, k, v = [torch.randn(B, T, h, D, generator=g) for h in (Nh, Nkv, Nkv)]
        mask = oops.build_packed_mask(doc)
        ql, kl, vl = [cp.shard(t).clone().requires_grad_() for t in (q, k, v)]
        events = []
        real_start = CP._start_forward

        def poisoned_start(cpar, locals_):
            fulls, tr = real_start(cpar, locals_)
            tr.wait()                                   # (gloo: land the data, then hide it again until `wait()`)
            stash = []
            mine = cpar.my_chunks()
            for f in fulls:
                for c in range(2 * cpar.cp):
                    if c not in mine:
                        view = f.narrow(1, c * cpar.Tc, cpar.Tc)
                        stash.append((view, view.clone()))
                        view.fill_(float("nan"))

            class Later:
                def wait(self_inner):
                    events.append("wait")
                    with torch.no_grad():
                        for view, data in stash:
                            view.copy_(data)
            return fulls, Later()
        CP._start_forward = poisoned_start
        with use_ops(oops):
            ex = CP.exchange_kv(cp, kl, vl)
            out = ex.attend(ql, mask, D ** -0.5)
            ref = oops.packed_attention_sharded(cp.shard(q), k, v, mask, cp.seq_shard(), D ** -0.5)
        ok_val = bool(torch.isfinite(out).all()) and float((out - ref).abs().max()) < 2e-5
        w = torch.randn(out.shape, generator=g)
        (out * w).sum().backward()
        ret[rank] = ("ok", ok_val, events, bool(torch.isfinite(ql.grad).all() and torch.isfinite(kl.grad).all()))
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_split_attention_reads_remote_chunks_only_behind_the_wait(world):
    """KVExchange.attend: the attention over the rank's OWN chunks must not touch a remote chunk (they are NaN until
    `wait()` here), the part over the received chunks runs behind the wait, and the LSE merge of the two equals the
    single attention over the global K/V (rows without any key in one part — e.g. a document that lives entirely in
    remote chunks — included)."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_split_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r][0] == "ok", results[r][1]
        assert results[r][1] and results[r][2] == ["wait"] and results[r][3], results[r]


def test_halo_return_travels_under_the_query_path_backward():
    """The overlap schedule of utils/context_parallel.py::exchange_kv in the backward pass: the partial dK/dV are SENT
    as soon as the attention backward has produced them, the query projection's backward runs next, and only then does
    the owner wait for them — right before the k/v projection backward that consumes them."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_order_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        results = dict(ret)
    for r in range(2):
        assert results[r][0] == "ok", results[r][1]
        assert results[r][1] == ["issue_return", "q_proj_backward", "finish_return", "k_proj_backward"], results[r][1]


# ------------------------------------------------------------------------------------------------ tensor parallel
KIMI_TINY = dict(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                 num_key_value_heads=2, head_dim=16, rms_norm_eps=1e-6, rope_theta=1e6, kimia_mimo_layers=1,
                 kimia_mimo_transformer_from_layer_index=1)


def _tp_batch():
    g = torch.Generator().manual_seed(4)
    B, T = 2, 48
    a, t = torch.randint(0, 64, (B, T), generator=g), torch.randint(0, 64, (B, T), generator=g)
    doc = torch.cat([torch.ones(B, 20), 2 * torch.ones(B, 20), torch.zeros(B, 8)], 1).long()
    pos = torch.cat([torch.arange(20), torch.arange(20), torch.zeros(8, dtype=torch.long)]).repeat(B, 1)
    labels = torch.where(doc > 0, torch.randint(0, 64, (B, T), generator=g), torch.full((B, T), -100))
    sl = torch.where(doc > 0, torch.full((B, T), 20), torch.ones(B, T, dtype=torch.long))
    return dict(text_input_ids=t, audio_input_ids=a, attention_mask=doc, position_ids=pos), labels, sl


def _tp_model():
    from touchnet_amd.models.kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM
    torch.manual_seed(21)
    m = KimiAudioPackedForCausalLM(KimiAudioConfig(**KIMI_TINY))
    m.post_init()
    for n, p in m.named_parameters():
        if n.endswith("bias"):
            torch.nn.init.normal_(p, std=0.05)
    return m


def _tp_worker(rank, world, port, ref_logits, ref_grads, ret, mode="plain"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle.ops as oops
        from torch.distributed.device_mesh import init_device_mesh
        from touchnet_amd.loss.cross_entropy import cross_entropy_loss
        from touchnet_amd.models.backend import use_ops
        from touchnet_amd.models.parallelize import parallelize_packed
        from touchnet_amd.utils.distributed import ParallelDims
        mesh = init_device_mesh("cpu", (world,), mesh_dim_names=("tp",))
        model = _tp_model()                                           # the SAME full weights on every rank ...
        job = types.SimpleNamespace(training_activation_checkpoint_mode="none", training_compile=False,
                                    training_enable_cpu_offload=False, training_tp_sequence_parallel=mode != "plain")
        dims = ParallelDims(1, 1, 1, world, 1, world, mode == "loss_parallel")
        model = parallelize_packed(model, mesh, dims, job)                                           # ... sharded here
        from touchnet_amd.models.tensor_parallel import reduce_sequence_partial_grads
        inputs, labels, sl = _tp_batch()
        with use_ops(oops):
            if mode == "loss_parallel":       # vocabulary-sharded head: only the fused lm_head + CE can evaluate it
                with pytest.raises(RuntimeError):
                    model(**inputs)
                out = model(**inputs, labels=labels, sentence_lens=sl, num_sentence=4)
                loss, err = out.loss, abs(float(out.loss) - float(ref_logits))     # (ref_logits carries the reference LOSS)
                assert 0.0 <= float(out.acc) <= 1.0
            else:
                out = model(**inputs)
                loss, _ = cross_entropy_loss(out.logits, labels, sl, 4)
                err = float((out.logits - ref_logits).abs().max())
            loss.backward()
        reduce_sequence_partial_grads(model)
        assert bool(model._tn_tp["seq_partial_names"]) == (mode != "plain")
        worst = 0.0
        sharded = model._tn_tp["sharded_names"]
        for n, p in model.named_parameters():
            if p.grad is None:                                        # mimo branch: not executed in training
                assert "mimo" in n, n
                continue
            ref = ref_grads[n]
            if n in sharded:
                dim = 1 if ("o_proj" in n or "down_proj" in n) else 0
                ref = ref.chunk(world, dim=dim)[rank]
            assert p.grad.shape == ref.shape, (n, p.grad.shape, ref.shape)
            worst = max(worst, float((p.grad - ref).abs().max()))
        ret[rank] = ("ok", err, worst, len(sharded))
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["plain", "sequence_parallel", "loss_parallel"])
def test_tensor_parallel_two_ranks_equals_single_process(mode):
    """Config E's TP = 2 plan on the Kimi-Audio decoder (heads and MLP columns split over 2 ranks) through
    `parallelize_fn(model, world_mesh, parallel_dims, job)`: logits identical to the unsharded model, gradients of
    tp-sharded parameters == the matching slices of the unsharded gradients, replicated parameters' gradients equal.
      plain              one all-reduce per attention / MLP output, mirrored in backward
      sequence_parallel  the reference's plan (parallelize_llama.py:133-176): residual stream and norms on T/tp rows,
                         all-gather in front of / reduce-scatter behind every block body, norm-weight gradients summed
                         over tp after the backward
      loss_parallel      + vocabulary-sharded lm_head inside the fused lm_head + CE (loss and every gradient, incl. the
                         head's shard, equal to the single-process ones)"""
    import oracle.ops as oops
    from touchnet_amd.loss.cross_entropy import cross_entropy_loss
    from touchnet_amd.models.backend import use_ops
    ref = _tp_model()
    inputs, labels, sl = _tp_batch()
    with use_ops(oops):
        out = ref(**inputs)
        loss, _ = cross_entropy_loss(out.logits, labels, sl, 4)
        loss.backward()
    ref_grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    target = loss.detach() if mode == "loss_parallel" else out.logits.detach()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_tp_worker, args=(2, _free_port(), target, ref_grads, ret, mode), nprocs=2, join=True)
        results = dict(ret)
    for r in range(2):
        assert results[r][0] == "ok", results[r][1]
        assert results[r][1] < 2e-5 and results[r][2] < 2e-5, results[r]
        # 3 + 1 mimo blocks x (7 weights + 3 biases) (+ the head's vocabulary shard)
        assert results[r][3] == 4 * 10 + (1 if mode == "loss_parallel" else 0)


def _tied_tp_worker(rank, world, port, ref_state, ref_loss, ref_grads, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle.ops as oops
        from torch.distributed.device_mesh import init_device_mesh
        from touchnet_amd.data.synthetic import text_batch
        from touchnet_amd.models.backend import use_ops
        from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
        from touchnet_amd.models.parallelize import parallelize_packed
        from touchnet_amd.models.tensor_parallel import reduce_sequence_partial_grads
        from touchnet_amd.utils.distributed import ParallelDims
        mesh = init_device_mesh("cpu", (world,), mesh_dim_names=("tp",))
        model = PackedCausalLM(DecoderConfig.from_dict(TINY))
        model.load_state_dict(ref_state)
        assert model.lm_head.weight is model.model.embed_tokens.weight
        job = types.SimpleNamespace(training_activation_checkpoint_mode="none", training_compile=False,
                                    training_enable_cpu_offload=False, training_tp_sequence_parallel=True)
        model = parallelize_packed(model, mesh, ParallelDims(1, 1, 1, world, 1, world, True), job)
        w = model.lm_head.weight
        assert w is model.model.embed_tokens.weight and w.shape[0] == TINY["vocab_size"] // world      # still ONE weight
        b = text_batch(16, 2, 32, seed=5, max_len=9)
        with use_ops(oops):
            out = model(input_ids=b["input_ids"], position_ids=b["position_ids"], attention_mask=b["attention_mask"],
                        labels=b["labels"], sentence_lens=b["sentence_lens"], num_sentence=b["num_sentence"])
            out.loss.backward()
        reduce_sequence_partial_grads(model)
        assert float(out.loss) == pytest.approx(ref_loss, rel=1e-5)
        worst = 0.0
        for n, p in model.named_parameters():
            ref = ref_grads[n]
            if n in model._tn_tp["sharded_names"]:
                ref = ref.chunk(world, dim=1 if ("o_proj" in n or "down_proj" in n) else 0)[rank]
            worst = max(worst, float((p.grad - ref).abs().max()))
        assert "model.embed_tokens.weight" in model._tn_tp["sharded_names"] and worst < 2e-5, worst
        ret[rank] = ("ok", worst)
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def test_tensor_parallel_with_tied_embeddings_and_loss_parallel():
    """Llama-3.2-1B / the reference's tiny test config tie lm_head to the embedding; with loss parallel the ONE weight is
    vocabulary-sharded and the embedding becomes vocabulary-parallel (the reference's RowwiseParallel on the embedding,
    parallelize_llama.py:133-141): loss and every gradient — the tied weight's rows included, embedding + head
    contributions — equal the single-process ones on 2 gloo ranks."""
    import oracle.ops as oops
    from touchnet_amd.data.synthetic import text_batch
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    torch.manual_seed(9)
    ref = PackedCausalLM(DecoderConfig.from_dict(TINY))
    ref.post_init()
    b = text_batch(16, 2, 32, seed=5, max_len=9)
    with use_ops(oops):
        out = ref(input_ids=b["input_ids"], position_ids=b["position_ids"], attention_mask=b["attention_mask"],
                  labels=b["labels"], sentence_lens=b["sentence_lens"], num_sentence=b["num_sentence"])
        out.loss.backward()
    state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    grads = {n: p.grad.clone() for n, p in ref.named_parameters()}
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_tied_tp_worker, args=(2, _free_port(), state, float(out.loss), grads, ret), nprocs=2, join=True)
        results = dict(ret)
    for r in range(2):
        assert results[r][0] == "ok", results[r][1]


def _group_mesh_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from touchnet_amd.utils.distributed import GroupMesh
        m = GroupMesh({"dp_shard": 2, "cp": 2}, {"dp": ("dp_replicate", "dp_shard"), "dp_shard_cp": ("dp_shard", "cp")})
        assert m.mesh_dim_names == ("dp_shard", "cp") and m["cp"].size() == 2 and m["dp_shard_cp"].size() == 4
        assert m["dp_shard"].get_local_rank() == rank // 2 and m["cp"].get_local_rank() == rank % 2
        assert m["dp_shard_cp"].get_local_rank() == rank and m["dp"].size() == 2
        # the groups are the ones a row-major DeviceMesh has: cp peers are neighbours, dp peers two apart
        t = torch.tensor([float(rank)])
        dist.all_reduce(t, group=m["cp"].get_group())
        u = torch.tensor([float(rank)])
        dist.all_reduce(u, group=m["dp_shard"].get_group())
        ret[rank] = ("ok", float(t), float(u))
    except Exception as e:
        import traceback
        ret[rank] = ("fail", traceback.format_exc(), repr(e))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_group_mesh_has_device_mesh_layout():
    """utils/distributed.GroupMesh (gloo groups for several ranks on one GPU) lays ranks out like torch's DeviceMesh."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_group_mesh_worker, args=(4, _free_port(), ret), nprocs=4, join=True)
        res = dict(ret)
    for r in range(4):
        assert res[r][0] == "ok", res[r][1]
        assert res[r][1] == float((r // 2) * 4 + 1) and res[r][2] == float(2 * (r % 2) + 2)


def _replica_check_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from touchnet_amd.utils.zero_dp import _check_replicas_drew_the_same_values as check
        g = torch.Generator().manual_seed(0)
        a, b = torch.randn(64, generator=g), torch.randn(7, 5, generator=g)
        out = {}
        check([a, b], dist.group.WORLD)                                   # same draw on both ranks: passes
        out["same"] = True
        # rank 1 holds a PERMUTATION of tensor 0 (same sum: the old single-total check let it through)
        try:
            check([a.flip(0) if rank == 1 else a, b], dist.group.WORLD)
            out["perm"] = "passed"
        except RuntimeError as e:
            out["perm"] = str(e)
        # compensating differences across two tensors (grand total unchanged)
        try:
            check([a + (1.0 / 64 if rank == 1 else 0.0), b - (1.0 / 35 if rank == 1 else 0.0)], dist.group.WORLD)
            out["comp"] = "passed"
        except RuntimeError as e:
            out["comp"] = str(e)
        if rank == 0:
            ret.update(out)
    finally:
        dist.destroy_process_group()


def test_flat_engine_replica_check_sees_permuted_and_compensating_draws():
    """the start-up check compared ONE float64 grand total over all parameters; a permuted tensor or differences
    that cancel passed, and a NaN raised the 'ranks differ' message.  Now per-tensor sums and sums of squares."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_replica_check_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        ret = dict(ret)
    assert ret["same"] is True
    assert "different initial parameter values" in ret["perm"] and "tensor 0" in ret["perm"]
    assert "different initial parameter values" in ret["comp"]
    from touchnet_amd.utils.zero_dp import _check_replicas_drew_the_same_values as check
    with pytest.raises(RuntimeError, match="non-finite"):
        check([torch.tensor([1.0, float("nan")])], None)
