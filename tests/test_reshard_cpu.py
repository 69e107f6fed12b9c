"""Twin of the reference's tests/touchnet/models/test_llama.py:85-150 for the MI355X spec: `parallelize_fn` builds the META
model under each (dp, cp, tp) mesh of the reference's parametrisation, the ranks load an UNSHARDED torch.distributed.checkpoint
into their shards and save it back sharded, one process reloads the sharded checkpoint — and gives the logits of the model
the first checkpoint was written from (abs 1e-6; same batch shape: 8 x 8 random ids, plain positions).  Same tiny
configuration as the reference's tests/assets/config/tiny_llama.json (d = 64); the CPU runs the oracle op set behind the
product modules, the sharding machinery (ParallelDims mesh, tensor-parallel plan incl. the vocabulary-parallel tied embedding,
FSDP2 over dp_shard x cp, the DTensor state_dict views of the tp shards — strided 2-D under tp x FSDP2) is the product's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.distributed.checkpoint as dcp
import torch.multiprocessing as mp

TINY = dict(vocab_size=16, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=8,
            num_key_value_heads=4, head_dim=8, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=True,
            initializer_range=0.02, max_position_embeddings=128,
            rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                              original_max_position_embeddings=64))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(meta: bool):
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.models.llama import DecoderConfig
    from touchnet_amd.utils.train_spec import get_train_spec
    spec = get_train_spec("llama_mi355")
    cfg = DecoderConfig.from_dict(TINY)
    if meta:
        with torch.device("meta"):
            return spec, spec.model_cls(cfg)
    return spec, spec.model_cls(cfg)


def _tiny_eval(rank, world, port, dims_kw, folder, shard_folder, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle.ops as oops
        from touchnet_amd.bin.train import TrainConfig
        from touchnet_amd.models.backend import use_ops
        from touchnet_amd.utils.distributed import ParallelDims
        dims = ParallelDims(**dims_kw)
        world_mesh = dims.build_mesh("cpu")
        spec, model = _build(meta=True)
        job = TrainConfig(training_model_name="llama_mi355", training_mixed_precision_param="float32",
                          training_dp_engine="fsdp2")
        with use_ops(oops):
            spec.parallelize_fn(model, world_mesh, dims, job)
            model.to_empty(device="cpu")
            with torch.no_grad():
                model.post_init()
                spec.additional_post_init_fn(model, torch.device("cpu"))
            model.eval()
            dcp.load({"model": model.state_dict()}, checkpoint_id=folder)          # unsharded checkpoint -> my shards
            dcp.save({"model": model.state_dict()}, checkpoint_id=shard_folder)    # ... and back out, sharded
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = traceback.format_exc() + repr(e)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dp,cp,tp", [(2, 1, 1, 2), (8, 8, 1, 1), (8, 2, 4, 1), (8, 4, 2, 1),
                                           (8, 2, 2, 2)])
def test_llama_reshards_through_parallelize_fn(tmp_path, world, dp, cp, tp):
    import oracle.ops as oops
    from touchnet_amd.models.backend import use_ops
    torch.manual_seed(0)
    _, model = _build(meta=False)
    with torch.no_grad():
        model.post_init()
    model.eval()
    folder, shard_folder = str(tmp_path / "step-0"), str(tmp_path / "step-0-sharded")
    dcp.save({"model": model.state_dict()}, checkpoint_id=folder, no_dist=True)
    ids = torch.randint(0, TINY["vocab_size"], (8, 8))
    pos = torch.arange(8).unsqueeze(0).repeat(8, 1)
    with use_ops(oops), torch.no_grad():
        want = model(input_ids=ids, position_ids=pos).logits.float().numpy()
    dims_kw = dict(dp_shard=dp, dp_replicate=1, cp=cp, tp=tp, pp=1, world_size=world, enable_loss_parallel=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_tiny_eval, args=(world, _free_port(), dims_kw, folder, shard_folder, ret), nprocs=world, join=True)
        results = dict(ret)
    for r in range(world):
        assert results[r] == "ok", results[r]
    spec, fresh = _build(meta=True)
    fresh.to_empty(device="cpu")
    with torch.no_grad():
        fresh.post_init()
        spec.additional_post_init_fn(fresh, torch.device("cpu"))
    dcp.load({"model": fresh.state_dict()}, checkpoint_id=shard_folder, no_dist=True)
    fresh.eval()
    with use_ops(oops), torch.no_grad():
        got = fresh(input_ids=ids, position_ids=pos).logits.float().numpy()
    assert got == pytest.approx(want, abs=1e-6)


def _tp_load_state_dict(rank, world, port, folder, full_path, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle.ops as oops
        from torch.distributed.tensor import DTensor
        from touchnet_amd.bin.train import TrainConfig
        from touchnet_amd.models.backend import use_ops
        from touchnet_amd.utils.distributed import ParallelDims
        dims = ParallelDims(dp_shard=1, dp_replicate=1, cp=1, tp=world, pp=1, world_size=world, enable_loss_parallel=True)
        world_mesh = dims.build_mesh("cpu")
        spec, model = _build(meta=True)
        job = TrainConfig(training_model_name="llama_mi355", training_mixed_precision_param="float32",
                          training_dp_engine="fsdp2")
        with use_ops(oops):
            spec.parallelize_fn(model, world_mesh, dims, job)
            model.to_empty(device="cpu")
            with torch.no_grad():
                model.post_init()
                spec.additional_post_init_fn(model, torch.device("cpu"))
            dcp.load({"model": model.state_dict()}, checkpoint_id=folder)
        want = {n: p.detach().clone() for n, p in model.named_parameters()}
        view = model.state_dict()
        assert any(isinstance(v, DTensor) for v in view.values())
        snapshot = {k: (DTensor.from_local(v.to_local().clone(), v.device_mesh, v.placements, run_check=False)
                        if isinstance(v, DTensor) else v.clone()) for k, v in view.items()}

        def scramble():
            with torch.no_grad():
                for p in model.parameters():
                    p.fill_(7.0)

        def check(tag):
            for n, p in model.named_parameters():
                assert not isinstance(p, DTensor) and p.shape == want[n].shape and torch.equal(p, want[n]), (tag, n)

        scramble()
        model.load_state_dict(snapshot)                      # (1) the model's own DTensor view
        check("view")
        scramble()
        full = torch.load(full_path)                         # (2) plain tensors of the GLOBAL shapes
        model.load_state_dict(full)
        check("full")
        scramble()
        model.load_state_dict({k: (v.to_local() if isinstance(v, DTensor) else v) for k, v in snapshot.items()})
        check("local")                                       # (3) plain tensors of the local shapes
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = traceback.format_exc() + repr(e)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_tensor_parallel_model_loads_its_own_state_dict_and_full_tensors(tmp_path):
    """`model.load_state_dict` under a real tp mesh (no DCP): the DTensor view `state_dict()` hands out, an unsharded
    state dict of plain tensors, and local-shaped plain tensors all end up as the local shards (tensor_parallel.py
    `_checkpoint_view`'s load hook; the reference's DTensor parameters take all three through DTensor.copy_ /
    torch.distributed.checkpoint.state_dict.set_model_state_dict)."""
    torch.manual_seed(0)
    _, model = _build(meta=False)
    with torch.no_grad():
        model.post_init()
    folder, full_path = str(tmp_path / "step-0"), str(tmp_path / "full.pt")
    dcp.save({"model": model.state_dict()}, checkpoint_id=folder, no_dist=True)
    torch.save(model.state_dict(), full_path)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_tp_load_state_dict, args=(2, _free_port(), folder, full_path, ret), nprocs=2, join=True)
        results = dict(ret)
    for r in range(2):
        assert results[r] == "ok", results[r]
