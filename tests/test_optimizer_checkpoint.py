"""FusedAdamW.state_dict() / load_state_dict() through torch.distributed.checkpoint — the way the reference's
CheckpointManager saves the optimizer (touchnet/utils/checkpoint.py).  DCP turns nested keys into strings and treats
plain tensors as replicated (saved once, reloaded everywhere): the state must be keyed by parameter name and, under
FSDP2, expose its local shards as DTensors.  Host logic only (no kernel runs): CPU, world size 1 and 2 (gloo)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.distributed.checkpoint as dcp
import torch.multiprocessing as mp
import torch.nn as nn

from touchnet_amd.utils.optimizer import FusedAdamW


def _net():
    torch.manual_seed(3)
    # 7 rows: shards over 2 ranks are uneven (4 + 3)
    return nn.Sequential(nn.Linear(5, 7, bias=True), nn.Linear(7, 3, bias=False))


def _fill(opt, salt):
    for i, s in enumerate(opt.state):
        for j, k in enumerate(("master", "m", "v")):
            s[k].copy_(torch.arange(s[k].numel(), dtype=torch.float32).view_as(s[k]) * (j + 1) + 100 * i + salt)
    opt.step_state[0] = 7.0


def test_dcp_round_trip_single_process(tmp_path):
    net = _net()
    opt = FusedAdamW(net.named_parameters(), lr=1e-3)
    assert opt.names == ["0.weight", "0.bias", "1.weight"]
    _fill(opt, 0.5)
    want = {n: {k: v.clone() for k, v in s.items()} for n, s in zip(opt.names, opt.state)}
    dcp.save({"optimizer": opt.state_dict()}, checkpoint_id=str(tmp_path / "ck"), no_dist=True)
    opt2 = FusedAdamW(_net().named_parameters(), lr=1e-3)
    sd = {"optimizer": opt2.state_dict()}
    dcp.load(sd, checkpoint_id=str(tmp_path / "ck"), no_dist=True)
    opt2.load_state_dict(sd["optimizer"])
    for n, s in zip(opt2.names, opt2.state):
        for k in ("master", "m", "v"):
            assert torch.equal(s[k], want[n][k]), (n, k)
    assert float(opt2.step_state[0]) == 7.0


def test_unnamed_parameters_get_string_keys_and_old_integer_keys_still_load():
    net = _net()
    opt = FusedAdamW(net.parameters())
    assert all(isinstance(k, str) for k in opt.state_dict()["state"])
    _fill(opt, 1.0)
    legacy = opt.state_dict()
    legacy["state"] = {i: v for i, v in enumerate(legacy["state"].values())}
    opt2 = FusedAdamW(_net().parameters())
    opt2.load_state_dict(legacy)
    assert torch.equal(opt2.state[2]["v"], opt.state[2]["v"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, path, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.distributed.device_mesh import init_device_mesh
        from torch.distributed.fsdp import fully_shard
        mesh = init_device_mesh("cpu", (world,))
        net = _net()
        fully_shard(net, mesh=mesh)
        opt = FusedAdamW(net.named_parameters(), lr=1e-3, process_group=mesh.get_group())
        shapes = [tuple(s["m"].shape) for s in opt.state]
        _fill(opt, 1000.0 * rank)                       # every rank's shard holds different numbers
        want = [{k: v.clone() for k, v in s.items()} for s in opt.state]
        sd = opt.state_dict()
        from torch.distributed.tensor import DTensor
        assert all(isinstance(v["exp_avg"], DTensor) for v in sd["state"].values())
        assert tuple(sd["state"]["0.weight"]["exp_avg"].shape) == (7, 5)          # the GLOBAL shape
        dcp.save({"optimizer": sd}, checkpoint_id=path)
        # a fresh optimizer over a fresh sharded model reloads ITS OWN shards
        net2 = _net()
        fully_shard(net2, mesh=mesh)
        opt2 = FusedAdamW(net2.named_parameters(), lr=1e-3, process_group=mesh.get_group())
        sd2 = {"optimizer": opt2.state_dict()}
        dcp.load(sd2, checkpoint_id=path)
        opt2.load_state_dict(sd2["optimizer"])
        ok = all(torch.equal(s[k], w[k]) for s, w in zip(opt2.state, want) for k in ("master", "m", "v"))
        ret[rank] = (ok, shapes)
    finally:
        dist.destroy_process_group()


def test_dcp_round_trip_two_rank_fsdp_uneven_shards(tmp_path):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path / "ck2"), ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0], dict(ret)
    assert ret[0][1][0] == (4, 5) and ret[1][1][0] == (3, 5), dict(ret)       # uneven shards were exercised


def _flat_worker(rank, world, port, path, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.distributed.device_mesh import init_device_mesh
        from torch.distributed.tensor import DTensor
        from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
        from touchnet_amd.utils.zero_dp import FlatShardedDataParallel
        mesh = init_device_mesh("cpu", (world,))
        cfg = DecoderConfig.from_dict(dict(vocab_size=16, hidden_size=64, intermediate_size=96, num_hidden_layers=2,
                                           num_attention_heads=8, num_key_value_heads=4, head_dim=8,
                                           tie_word_embeddings=False))

        def build():
            torch.manual_seed(5)
            m = PackedCausalLM(cfg)
            eng = FlatShardedDataParallel(m, mesh)
            return eng, FusedAdamW(eng.named_shards(), lr=1e-3, process_group=mesh.get_group())
        eng, opt = build()
        assert opt.names[:2] == ["block.0.0", "block.0.1"]
        _fill(opt, 1000.0 * rank)
        want = [{k: v.clone() for k, v in s.items()} for s in opt.state]
        sd = opt.state_dict()
        first = sd["state"]["block.0.0"]["exp_avg"]
        assert isinstance(first, DTensor) and tuple(first.shape) == (eng.buckets[0].total,)      # the whole flat buffer
        assert first.to_local().numel() == eng.buckets[0].S
        dcp.save({"optimizer": sd}, checkpoint_id=path)
        eng2, opt2 = build()
        sd2 = {"optimizer": opt2.state_dict()}
        dcp.load(sd2, checkpoint_id=path)
        opt2.load_state_dict(sd2["optimizer"])
        ok = all(torch.equal(s[k], w[k]) for s, w in zip(opt2.state, want) for k in ("master", "m", "v"))
        # what the other world size must find: every rank's slices, in rank order
        ret[rank] = (ok, [eng.buckets[i].total for i in range(len(eng.buckets))],
                     [{k: v.clone() for k, v in s.items()} for s in want])
    finally:
        dist.destroy_process_group()


def test_dcp_round_trip_of_the_flat_engines_shards(tmp_path):
    """utils/zero_dp.py: the optimizer state of a bucket is this rank's slice of a flat buffer; it is saved as a dim-0
    sharded DTensor of the whole buffer and every rank reloads its own slice."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, str(tmp_path / "ck3"), ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0], dict(ret)
    # ---- the checkpoint written by TWO ranks is read by ONE: bucket lengths do not depend on the world size
    # (worlds dividing 64), so DCP reshards the 1-D DTensors; the lone rank holds rank 0's slice followed by rank 1's
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from torch.distributed.device_mesh import init_device_mesh
        from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
        from touchnet_amd.utils.zero_dp import FlatShardedDataParallel
        cfg = DecoderConfig.from_dict(dict(vocab_size=16, hidden_size=64, intermediate_size=96, num_hidden_layers=2,
                                           num_attention_heads=8, num_key_value_heads=4, head_dim=8,
                                           tie_word_embeddings=False))
        torch.manual_seed(5)
        mesh = init_device_mesh("cpu", (1,))
        eng = FlatShardedDataParallel(PackedCausalLM(cfg), mesh)
        opt = FusedAdamW(eng.named_shards(), lr=1e-3, process_group=mesh.get_group())
        assert [b.total for b in eng.buckets] == ret[0][1] == ret[1][1]
        sd = {"optimizer": opt.state_dict()}
        dcp.load(sd, checkpoint_id=str(tmp_path / "ck3"))
        opt.load_state_dict(sd["optimizer"])
        for i, st in enumerate(opt.state):
            for k in ("master", "m", "v"):
                assert torch.equal(st[k], torch.cat([ret[0][2][i][k], ret[1][2][i][k]])), (i, k)
    finally:
        dist.destroy_process_group()
